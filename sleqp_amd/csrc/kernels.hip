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
// dependency wait, and three single-launch dataflow kernels (k_factor_top, k_fwd_top,
// k_bwd_top) in which workgroups synchronise through counters (DESIGN.md sections 2 and 4).
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
// PACKED: one word per product, (min(a, b) << 8) | |a - b| — both factors sit in the same column
// of K, a few entries apart, so the pair usually fits (halves the index traffic, which is what this
// kernel moves: its K values stay in L2).
template <bool PACKED>
__device__ __forceinline__ void prod_pair(const int* __restrict__ prod_a, const int* __restrict__ prod_b, long long p,
                                          int& a, int& b) {
  if (PACKED) {
    const unsigned int pk = (unsigned int)prod_a[p];
    a = (int)(pk >> 8);
    b = a + (int)(pk & 255u);
  } else {
    a = prod_a[p];
    b = prod_b[p];
  }
}

// One wave owns 64 consecutive entries and with them one contiguous range of the product list.
// The lists are short on average (two products) but every column has a diagonal entry with one
// product per nonzero of its row of A: a lane walking its own list serialises that many dependent
// index -> value round trips while the other 63 lanes idle, and nearly every wave holds such a
// lane.  So the wave loads its whole range cooperatively (coalesced index words, every gather
// useful, 2 * MV_U independent loads in flight per lane), parks the factor pairs in LDS, and each
// lane then sums its own products from there in list order (same fma chain, same bits).
constexpr int MV_U = 4;
template <class IDX, bool PACKED>
__global__ __launch_bounds__(FB) void k_mvals_prod(long long nM, const IDX* __restrict__ prod_ptr,
                                                   const int* __restrict__ prod_a, const int* __restrict__ prod_b,
                                                   const IDX* __restrict__ target, const double* __restrict__ Kval,
                                                   double* __restrict__ L, long long ng, int nbg,
                                                   const int* __restrict__ gsrc, double* __restrict__ gout) {
  // the first nbg blocks carry an unrelated small job along (values of A in pivot order for the
  // solves' SpMVs, gout[i] = Kval[gsrc[i]]): one launch and its ramp less on the critical path
  if ((int)blockIdx.x < nbg) {
    for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < ng; i += (long long)nbg * FB) gout[i] = Kval[gsrc[i]];
    return;
  }
  constexpr int CH = 64 * MV_U;
  __shared__ double2 pairs[FB / 64][CH];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double2* buf = pairs[wv];
  const long long bid = blockIdx.x - nbg, nbm = gridDim.x - nbg;
  for (long long e0 = bid * (long long)FB + wv * 64; e0 < nM; e0 += nbm * FB) {
    const long long e = e0 + lane;
    const bool valid = e < nM;
    const long long eend = (e0 + 64 < nM) ? e0 + 64 : nM;
    const long long P0 = (long long)prod_ptr[e0], P1 = (long long)prod_ptr[eend];  // wave-uniform
    const long long p0 = valid ? (long long)prod_ptr[e] : P1, p1 = valid ? (long long)prod_ptr[e + 1] : P1;
    const long long tgt = valid ? (long long)target[e] : 0;
    double s = 0.0;
    for (long long c = P0; c < P1; c += CH) {
      int a[MV_U], b[MV_U];
#pragma unroll
      for (int u = 0; u < MV_U; ++u) {
        const long long p = c + u * 64 + lane;
        prod_pair<PACKED>(prod_a, prod_b, p < P1 ? p : P1 - 1, a[u], b[u]);
        if (p >= P1) a[u] = b[u] = 0;  // unused slots share one cache line
      }
      double2 v[MV_U];
#pragma unroll
      for (int u = 0; u < MV_U; ++u) {
        v[u].x = Kval[a[u]];
        v[u].y = Kval[b[u]];
      }
#pragma unroll
      for (int u = 0; u < MV_U; ++u) buf[u * 64 + lane] = v[u];
      // LDS operations of one wave execute in order: no workgroup barrier, only keep the compiler
      // from moving the reads above the writes
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int lo = (int)((p0 > c ? p0 : c) - c);
      const int hi = (int)((p1 < c + CH ? p1 : c + CH) - c);
      for (int q = lo; q < hi; ++q) {
        const double2 f = buf[q];
        s = fma(f.x, f.y, s);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (valid) L[tgt] = s;
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
// Front factorisation.  Every dense phase runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64, verified lane layout: A[l&15][l>>4], B[l>>4][l&15],
// D[(l>>4)+4q][l&15]).
//
//   A   (scatter form, k_front_assemble: fused small-front levels and cross-check only) zero the
//       lower triangle of the update matrix U_s and extend-add the children's update matrices
//       (relative indices), partitioned by TARGET column class so that several workgroups can
//       assemble one front without write conflicts and in a fixed order.  The split kernels
//       below PULL instead: each gathers the children's entries of exactly what it is about to
//       use (inverse relative indices, child order => same bits as the scatter form).
//   B   blocked LDL^T (nb = 16) of the w x w pivot block in LDS:
//         S1  16x16 diagonal block: LDL^T and its inverse in registers of one wave; every lane
//             of a 16-lane row holds a whole matrix row, so a pivot step is one reciprocal and a
//             run of v_fmac_f64 with a DPP row broadcast on their first source (no LDS at all)
//         S2  block column  L_Ik = A_Ik inv(L_kk)^T D^-1          (MFMA)
//         S3  trailing update A_IJ -= L_Ik D L_Jk^T                (MFMA), look-ahead: wave 0
//             factors the next diagonal block meanwhile, and the other waves also form the
//             inverse of the unit lower factor block row by block row (so that the solves are
//             pure GEMVs) and store finished tiles
//   C   L21 = P21 inv(L11)^T D^-1, panel rows streamed, X from LDS   (MFMA)
//   D   U_s = sum(children) - L21 D L21^T in 64 x 64 tiles, operand strips staged in LDS
//       (k-major, 32-pivot chunks), 32 x 32 block per wave           (MFMA)
//
// Levels of tiny fronts run the fused kernel (one workgroup per front, phases B-D back to
// back after a scatter assembly); wide levels run one kernel per phase with many workgroups per
// front (k_front_pivot / _panel / _schur); the narrow top of the tree runs as ONE dataflow
// launch (k_factor_top).
//
// LDS layout: dd[wp] | region.  Phase B: region = A (wp x lda, lda = wp + 1: row
// fragments and transposed fragments are both <= 2-way bank conflicted) followed
// by the 16-column panel Y.  Phase D: region = two 64 x KC operand strips.
// ---------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));
// In-kernel timeline (scripts/timeline.py builds a copy of the library with -DHIPFACT_TRACE): wall_clock64 stamps
// of every workgroup of the dataflow launch (start, dependency met, work done, published) and of the chain wave
// of its pivot role.  Compiled out of the product.
#ifdef HIPFACT_TRACE
constexpr int TRACE_WGS = 8192;
__device__ long long g_trace[TRACE_WGS * 8];
__device__ long long g_piv[TRACE_WGS * 24];
__device__ long long g_own[TRACE_WGS * 8];  // row I of the pivot block handed to the chain wave (its owner's last update)
#define TRW(slot)                                                                                  \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < TRACE_WGS) g_trace[blockIdx.x * 8 + (slot)] = wall_clock64(); \
  } while (0)
#define TRP(slot)                                                                                \
  do {                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < TRACE_WGS) g_piv[blockIdx.x * 24 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define TRW(slot) \
  do {            \
  } while (0)
#define TRP(slot) \
  do {            \
  } while (0)
#endif
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// 1/d to ~1 ulp without the IEEE division sequence.  v_rcp_f64 is good to 2^-24.4 on gfx950
// (scripts/probe/rcp_f64_precision.hip), so ONE cubic step x0 (1 + e + e^2), e = 1 - d x0, leaves
// an error of e^3 < 2^-73 before rounding: three dependent FMAs instead of the four of two
// Newton steps.  Sits on the sequential pivot chain of the diagonal blocks.
__device__ __forceinline__ double fast_rcp(double d) {
  const double x = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, x, 1.0);
  const double t = fma(e, e, e);
  return fma(x, t, x);
}

// ---- data as its own flag (single-launch solve sweeps, pivot -> panel in the single-launch
// factorisation).  A dependent hop through a flag costs
// three memory round trips in a row: the producer's release, the consumer's poll seeing the flag,
// then the consumer's loads of the data.  The vectors exchanged between fronts are small, so the
// ordinary fronts exchange them element by element instead: every element is stored with an
// agent-scope atomic store and the consumer polls the element itself until it no longer holds
// the sentinel (all bits set: a NaN that no arithmetic produces) - one round trip.  The slots are
// put back to the sentinel by the opposite sweep (update vectors: backward item of the front;
// solution copy ysol: forward item of the front), i.e. in a different launch.
constexpr unsigned long long SOLVE_SENT = ~0ull;
__device__ __forceinline__ double poll_f64(const double* __restrict__ p, int* __restrict__ info) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  unsigned long long bits;
  int spins = 0;
  while ((bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == SOLVE_SENT) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 20)) {
      atomicAdd(&info[INFO_TIMEOUT], 1);
      break;
    }
  }
  return __longlong_as_double((long long)bits);
}
__device__ __forceinline__ void post_f64(double* __restrict__ p, double v) {
  // a NaN that carries the sentinel's payload (a caller's right-hand side may hold any bit pattern, and
  // NaN payloads propagate through arithmetic) is posted as the canonical quiet NaN instead
  unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  bits = (bits == SOLVE_SENT) ? 0x7FF8000000000000ull : bits;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sent_f64(double* __restrict__ p) {
  *reinterpret_cast<unsigned long long*>(p) = SOLVE_SENT;
}
// Same, for a slot that is posted or polled again inside the SAME launch.  The eight XCDs have an L2
// each; a plain store may sit there as a dirty line and be written back over a value that another XCD
// has posted in the meantime with an agent-scope store.  (Resets whose next use is a launch away may
// stay plain: the end of a kernel writes the L2s back.)
__device__ __forceinline__ void sent_f64_agent(double* __restrict__ p) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), SOLVE_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FrontCtx {
  int w, r, u, wp, nbk, lda;
  double* P;   // panel (global)
  double* Us;  // update matrix (global)
  double* dd;  // pivots (LDS)
  double* A;   // pivot block / inverse (LDS)
  double* Yp;  // block-column panel (LDS)
  double* Xa;  // single-launch factorisation: where the finished tiles of inv(L11) are posted for the panel workgroups (else null)
};

template <class Desc>
__device__ __forceinline__ FrontCtx make_ctx(const Desc& S, double* L, double* U, double* lds) {
  FrontCtx c;
  c.w = S.w;
  c.r = S.r;
  c.u = S.r - S.w;
  c.wp = (S.w + 15) & ~15;
  c.nbk = c.wp >> 4;
  c.lda = c.wp + 1;
  c.P = L + S.Loff;
  c.Us = U + S.Uoff;
  c.dd = lds;
  c.A = lds + c.wp;
  c.Yp = c.A + c.wp * c.lda;
  c.Xa = nullptr;
  return c;
}

// ---- pull-mode extend-add (split kernels).  Instead of a separate assembly pass
// that scatters the children's update matrices into the parent (phase A), the
// phases B / C / D of the parent gather the contributions of element (i, j)
// themselves: inv_c[p] = row of child c's update matrix that maps to front row p
// (-1: none).  Children are added in child order on top of the original entry,
// which is exactly the order of dev_assemble: both paths give identical bits.
struct PullCtx {
  int n;
  const double* Uc[MAXCH];
  const int* inv[MAXCH];
  const int* rel[MAXCH];
  int uc[MAXCH];
  int next;  // further children of the front: block index in the overflow array (fronts with > MAXCH children)
};

__device__ __forceinline__ PullCtx make_pull(const PullDesc& D, const double* __restrict__ U,
                                             const int* __restrict__ inv, const int* __restrict__ rel, int pull) {
  PullCtx pc;
  pc.n = pull ? D.n : 0;
  pc.next = pull ? D.next : -1;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) {
    pc.Uc[ch] = nullptr;
    pc.inv[ch] = nullptr;
    pc.rel[ch] = nullptr;
    pc.uc[ch] = 0;
    if (ch < pc.n) {
      pc.Uc[ch] = U + D.Uoff[ch];
      pc.inv[ch] = inv + D.invoff[ch];
      pc.rel[ch] = rel + D.reloff[ch];
      pc.uc[ch] = D.uc[ch];
    }
  }
  return pc;
}

// ---- phase A (any block size).  Fronts without children are not assembled at
// all: their Schur update is written in assign mode (phase D).  `relbuf` is an
// LDS int array of max(u_child) entries: the child's relative indices are staged
// once, so that the class test and the scatter addresses cost no dependent
// global round trips.
__device__ __forceinline__ void dev_assemble(const SnDesc& S, const FrontCtx& c, const SnDesc* __restrict__ sn,
                                             const double* __restrict__ U, const int* __restrict__ rel,
                                             const int* __restrict__ child_idx, int part, int nparts,
                                             int* relbuf) {
  if (S.child_begin == S.child_end) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = blockDim.x >> 6;
  const int w = c.w, r = c.r, u = c.u;
  // zero the lower triangle of the U columns of this class
  for (int jj = wave; jj < u; jj += nw) {
    if (nparts > 1 && ((jj + w) % nparts) != part) continue;
    double* colp = c.Us + (long long)jj * u;
    for (int i = jj + lane; i < u; i += 64) colp[i] = 0.0;
  }
  for (int ci = S.child_begin; ci < S.child_end; ++ci) {
    const SnDesc Cd = sn[child_idx[ci]];
    const int uc = Cd.r - Cd.w;
    const double* __restrict__ Uc = U + Cd.Uoff;
    const int* __restrict__ rc = rel + Cd.reloff;
    __syncthreads();  // zeroing / previous child finished; relbuf free
    for (int a = tid; a < uc; a += blockDim.x) relbuf[a] = rc[a];
    __syncthreads();
    for (int b = wave; b < uc; b += nw) {
      const int tb = relbuf[b];
      if (nparts > 1 && (tb % nparts) != part) continue;
      const double* __restrict__ col = Uc + (long long)b * uc;
      double* dst = (tb < w) ? c.P + (long long)tb * r : c.Us + (long long)(tb - w) * u - w;
      for (int a0 = b; a0 < uc; a0 += 256) {
        int t[4];
        double v[4], o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = a0 + lane + 64 * q;
          t[q] = (a < uc) ? relbuf[a] : -1;
          v[q] = (a < uc) ? col[a] : 0.0;
          o[q] = (a < uc) ? dst[t[q]] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (t[q] >= 0) dst[t[q]] = o[q] + v[q];
      }
    }
  }
}

// ---- 16 x 16 diagonal block kb: LDL^T and the inverse of its unit lower factor in the registers
// of ONE wave.  Leaves inv(L_kk) in the block and the pivots in dd.
// Lane exchange inside the 16-lane rows through the data-parallel path of the vector ALU.
// gfx90a+ allows exactly one DPP control for 64-bit operands, row_newbcast:K (every lane reads
// lane K of its row), and it can sit on the first source of v_fmac_f64: broadcast and FMA are ONE
// instruction of ~6.5 ns, where two ds_swizzle + FMA cost 12 ns to issue and 28 ns on a dependent
// chain (scripts/probe/lane_exchange_latency.hip).  The compiler does not form the fused
// instruction by itself, hence the assembly.  Hazard: the hardware needs two wait states between a
// vector write of a register and its use as a DPP source, and the hazard recogniser does not look
// into inline assembly.  The statements are volatile, i.e. they stay in source order, and the source
// order guarantees the distance for the FMAs: every register they broadcast was written by the
// previous elimination step, at least four of these statements earlier (the x updates close every
// step, behind an s_nop of their own).  Only the broadcast of the next pivot follows its producer directly; it carries
// its own s_nop.  (An s_nop in front of every FMA costs 4 % of the whole factorisation.)
// tests/test_isa_hazards.py checks the distances in the compiled code.
template <int K>
__device__ __forceinline__ double rowb_f64(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
// acc += (lane K of the row: src) * mul
template <int K>
__device__ __forceinline__ void fmac_rowb_f64(double& acc, double src, double mul) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mul), "n"(K));
}

// Elimination step K of the diagonal block.  Lane (i = li, q = lk) holds the WHOLE row i of the
// (symmetric) block in a[0..15], replicated over q, and X[i][4 cc + q] in x[cc]: the multiplier
// l_ik = A[i][k] / d_k is then a lane-local product and row k arrives by row broadcasts only -
// nothing crosses the 16-lane rows, nothing goes through LDS.  X[k][j] = 0 for j > k, so the
// slices cc > k / 4 of x are not touched yet.  nl is -l_ik of THIS step, computed at the end
// of the previous one; the next pivot column is updated first.
// Rows i <= k are finished: their multiplier must be zero for x (its rows are the result), but
// their part of a is never read again (row k was broadcast in this very step, the pivots are
// captured when they become final), so the a updates take the raw multiplier and the select
// stays off the dependent chain.
#ifndef HIPFACT_DIAG_INTERLEAVED
template <int K>
__device__ __forceinline__ double diag_step(double (&a)[16], double (&x)[4], double& dsel, int li, double nl) {
  double nl_next = 0.0;
  if (K < 15) fmac_rowb_f64<K>(a[K + 1], a[K + 1], nl);
  dsel = (li == K + 1) ? a[K + 1] : dsel;  // pivot k + 1 is final now
  if (K < 14) {
    const double d = rowb_f64<K + 1>(a[K + 1]);
    nl_next = -a[K + 1] * fast_rcp(d);
  }
#pragma unroll
  for (int j = K + 2; j < 16; ++j) fmac_rowb_f64<K>(a[j], a[j], nl);  // columns j <= k are dead
  const double nlx = (li > K) ? nl : 0.0;
  // (a slice of x that is touched for the first time was initialised by compiler-generated vector
  // code, and nlx is selected right here: two wait states, once per step)
  asm volatile("s_nop 1");
#pragma unroll
  for (int cc = 0; cc < 4; ++cc)
    if (4 * cc <= K) fmac_rowb_f64<K>(x[cc], x[cc], nlx);
  return nl_next;
}

#else
// The wave issues in order, and every instruction of the pivot chain (update of the next pivot -> broadcast ->
// reciprocal -> cubic correction, three FMAs -> multiplier) waits for its predecessor.  Left to the compiler the
// chain ends up in one piece behind the row updates of the step (212 clocks per pivot: chain + updates); here
// the updates of the step (a[K+2..15], then the slices of x) are dealt between the chain instructions, so that
// they issue in the shadow of its latencies.  Everything is volatile assembly, i.e. stays in this order; the
// reciprocal sequence is that of fast_rcp, operation by operation (same bits).  Hazards the recogniser cannot
// see inside assembly: a DPP source needs two wait states after its producer (the chain's broadcast carries an
// s_nop, the updates read registers of the previous step), a transcendental result one before its use.
template <int K, int IDX>
__device__ __forceinline__ void diag_bulk(double (&a)[16], double (&x)[4], double nl, double nlx) {
  constexpr int NA = (K < 14) ? 14 - K : 0;  // a[K+2 .. 15]; columns j <= k are dead
  if constexpr (IDX < NA) {
    fmac_rowb_f64<K>(a[K + 2 + IDX], a[K + 2 + IDX], nl);
  } else if constexpr (IDX - NA < 4 && 4 * (IDX - NA) <= K) {
    // (a slice of x that is touched for the first time was initialised by compiler-generated vector code, and
    // nlx is selected by it as well: two wait states, once per step)
    if constexpr (IDX == NA) asm volatile("s_nop 1");
    fmac_rowb_f64<K>(x[IDX - NA], x[IDX - NA], nlx);
  }
}
template <int K>
__device__ __forceinline__ double diag_step(double (&a)[16], double (&x)[4], double& dsel, int li, double nl) {
  double nl_next = 0.0;
  const double nlx = (li > K) ? nl : 0.0;  // rows i <= k of x are finished
  if (K < 15) fmac_rowb_f64<K>(a[K + 1], a[K + 1], nl);
  diag_bulk<K, 0>(a, x, nl, nlx);
  if constexpr (K < 14) {
    double d, x0, e, t, rc;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                 : "=v"(d)
                 : "v"(a[K + 1]), "n"(K + 1));
    diag_bulk<K, 1>(a, x, nl, nlx);
    asm volatile("v_rcp_f64 %0, %1" : "=v"(x0) : "v"(d));
    diag_bulk<K, 2>(a, x, nl, nlx);
    diag_bulk<K, 3>(a, x, nl, nlx);
    diag_bulk<K, 4>(a, x, nl, nlx);
    asm volatile("s_nop 0\n\tv_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(d), "v"(x0));
    diag_bulk<K, 5>(a, x, nl, nlx);
    diag_bulk<K, 6>(a, x, nl, nlx);
    asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(t) : "v"(e));
    diag_bulk<K, 7>(a, x, nl, nlx);
    diag_bulk<K, 8>(a, x, nl, nlx);
    asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(rc) : "v"(x0), "v"(t));
    diag_bulk<K, 9>(a, x, nl, nlx);
    diag_bulk<K, 10>(a, x, nl, nlx);
    asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nl_next) : "v"(a[K + 1]), "v"(rc));
    diag_bulk<K, 11>(a, x, nl, nlx);
    diag_bulk<K, 12>(a, x, nl, nlx);
    diag_bulk<K, 13>(a, x, nl, nlx);
    diag_bulk<K, 14>(a, x, nl, nlx);
    diag_bulk<K, 15>(a, x, nl, nlx);
    diag_bulk<K, 16>(a, x, nl, nlx);
    diag_bulk<K, 17>(a, x, nl, nlx);
  } else {
    diag_bulk<K, 1>(a, x, nl, nlx);
    diag_bulk<K, 2>(a, x, nl, nlx);
    diag_bulk<K, 3>(a, x, nl, nlx);
    diag_bulk<K, 4>(a, x, nl, nlx);
  }
  dsel = (li == K + 1) ? a[K + 1] : dsel;  // pivot k + 1 is final since the first instruction of the step
  return nl_next;
}
#endif

// xout: the inverse as the lane holds it, X[li][4 cc + lk] - the A operand layout of the matrix cores (V2 chain)
__device__ __forceinline__ void dev_diag_block(const FrontCtx& c, double* scratch, int k0, int* __restrict__ info,
                                               double* xout = nullptr) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int lda = c.lda;
  double* A = c.A;
  double a[16], x[4];
  // full symmetric row from the stored lower triangle.  ONE load per element: two loads under a ternary are not
  // speculated and a selected index a * lda + b is not if-converted either - both become sixteen exec-mask
  // branches around a quarter-rate 64-bit multiply-add (900 of the 3400 clocks of a diagonal block,
  // scripts/timeline.py).
  // Element j of row li lives at (li, j) for j <= li and at (j, li) beyond: a walk along the row up to the diagonal
  // and down the column from there, i.e. a running index with a selected increment (no multiply, no branch).
  int at = (k0 + li) + k0 * lda;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    a[j] = A[at];
    at += (j < li) ? lda : 1;
  }
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) x[cc] = (li == 4 * cc + lk) ? 1.0 : 0.0;
  // The dependent chain of the whole front runs through these steps (pivot -> reciprocal ->
  // multiplier -> update of the next pivot column): no selects on the chain; singular /
  // negative pivots are counted afterwards (a zero pivot floods the block with non-finite
  // values; the factorisation is reported singular either way).
  double dsel = a[0];  // pivot 0 (lane li = 0 keeps it)
  double nl;
  {
    const double d = rowb_f64<0>(a[0]);
    nl = -a[0] * fast_rcp(d);
  }
  nl = diag_step<0>(a, x, dsel, li, nl);
  nl = diag_step<1>(a, x, dsel, li, nl);
  nl = diag_step<2>(a, x, dsel, li, nl);
  nl = diag_step<3>(a, x, dsel, li, nl);
  nl = diag_step<4>(a, x, dsel, li, nl);
  nl = diag_step<5>(a, x, dsel, li, nl);
  nl = diag_step<6>(a, x, dsel, li, nl);
  nl = diag_step<7>(a, x, dsel, li, nl);
  nl = diag_step<8>(a, x, dsel, li, nl);
  nl = diag_step<9>(a, x, dsel, li, nl);
  nl = diag_step<10>(a, x, dsel, li, nl);
  nl = diag_step<11>(a, x, dsel, li, nl);
  nl = diag_step<12>(a, x, dsel, li, nl);
  nl = diag_step<13>(a, x, dsel, li, nl);
  nl = diag_step<14>(a, x, dsel, li, nl);
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) A[(k0 + li) + (k0 + 4 * cc + lk) * lda] = x[cc];
  if (xout) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) xout[cc] = x[cc];
  }
  const bool owner = (lk == 0);
  const bool bad = owner && ((dsel == 0.0) || !(fabs(dsel) <= 1.7e308));  // exactly singular or non-finite
  const bool neg = owner && !bad && (dsel < 0.0);
  if (owner) {
    c.dd[k0 + li] = bad ? 1.0 : dsel;
    scratch[li] = fast_rcp(bad ? 1.0 : dsel);  // 1 / d for the block column (S2) of this step
  }
  const int nzero = __popcll(__ballot(bad)), nneg = __popcll(__ballot(neg));
  if (lane == 0 && (nzero | nneg)) {
    if (nzero) atomicAdd(&info[INFO_ZERO_PIVOT], nzero);
    if (nneg) atomicAdd(&info[INFO_NEG_PIVOT], nneg);
  }
}

// one 16 x 16 tile (I, J) of the trailing update A_IJ -= L_Ik Y_Jk^T (MFMA)
__device__ __forceinline__ void dev_trailing_tile(const FrontCtx& c, int k0, int I, int J) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int lda = c.lda;
  double* A = c.A;
  const double* Yp = c.Yp;
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

// ---- phase B: pivot block in LDS -> inv(L11) (unit lower) in A, pivots in dd.
// Blocked right-looking LDL^T (nb = 16) with look-ahead: in the trailing update of
// step kb, wave 0 updates the next diagonal tile first and factors it at once,
// while the other waves finish the remaining tiles.  Any number of waves >= 1.
// overflow blocks of children descriptors (fronts with more than MAXCH children) and what is needed to resolve them
struct PullMore {
  const PullDesc* more;
  const double* U;
  const int* inv;
  const int* rel;
};
__device__ __forceinline__ PullMore no_more() { return PullMore{nullptr, nullptr, nullptr, nullptr}; }

// pivot-block gathers of one block of children (maps of the pivot rows in LDS: invb), added in child order
__device__ __forceinline__ void pivot_gather_stage1(const PullCtx& pc, const int* invb, int wp, int i1, int kq1,
                                                    double (&v1)[4]) {
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (ch < pc.n) {
      const double* __restrict__ Uc = pc.Uc[ch];
      const int uc = pc.uc[ch];
      const int* ib = invb + ch * wp;
      const int ci = (i1 < wp) ? ib[i1] : -1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ck = ib[kq1 + 4 * q];
        const bool ok = ci >= 0 && ck >= 0 && ci >= ck;
        const double g = Uc[ok ? ci + (long long)ck * uc : 0];
        v1[q] += ok ? g : 0.0;
      }
    }
}
// The same gathers in two halves for the children that are awaited (k_factor_top): everything up to the
// offsets inside the children's update matrices (maps, index arithmetic; they fit 32 bits: uc <= 2^15) happens
// BEFORE the wait, so that behind it only the loads themselves remain - all of them in flight at once.
__device__ __forceinline__ void pivot_offsets_stage1(const PullCtx& pc, const int* invb, int wp, int i1, int kq1,
                                                     int (&o1)[MAXCH][4]) {
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) {
    const int* ib = invb + ch * wp;
    const int uc = pc.uc[ch];
    const int ci = (ch < pc.n && i1 < wp) ? ib[i1] : -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ck = (ch < pc.n) ? ib[kq1 + 4 * q] : -1;
      o1[ch][q] = (ci >= 0 && ck >= 0 && ci >= ck) ? ci + ck * uc : -1;
    }
  }
}
// Stage 2 of the pivot-block load (waves 1..7): the lower triangle of the columns 16 .. wp-1 only.
// Column 16 + p is folded with column wp - 1 - p, so that every pair holds wp - 15 entries: slot
// (s, j) of a thread is entry t = lane + 64 s of pair p = (wave - 1) + 7 j.
__device__ __forceinline__ bool stage2_elem(int wp, int wave, int lane, int s, int j, int& i, int& k) {
  const int npairs = (wp - 16) >> 1;
  const int p = (wave - 1) + 7 * j;
  const int t = lane + 64 * s;
  const int ka = 16 + p, kb = wp - 1 - p;
  const int len1 = wp - ka;
  const bool first = t < len1;
  k = first ? ka : kb;
  i = first ? ka + t : kb + (t - len1);
  return wave >= 1 && p < npairs && t < wp - 15;
}
__device__ __forceinline__ void pivot_gather_stage2_from(const PullCtx& pc, const int* invb, int wp, int lane, int wave,
                                                         double (&v)[2][8], int cfirst) {
  // two children in flight at a time, added in child order
#pragma unroll
  for (int c0 = 0; c0 < MAXCH; c0 += 2)
    if (c0 >= cfirst && c0 < pc.n) {
      double g[2][2][8];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int ch = c0 + cc;
        const bool has = ch < pc.n;
        const double* __restrict__ Uc = has ? pc.Uc[ch] : pc.Uc[c0];
        const int uc = has ? pc.uc[ch] : 0;
        const int* ib = invb + (has ? ch : c0) * wp;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            int i, k;
            const bool valid = stage2_elem(wp, wave, lane, s2, j, i, k);
            const int ci = (valid && has) ? ib[i] : -1;
            const int ck = (valid && has) ? ib[k] : -1;
            const bool ok = ci >= 0 && ck >= 0;  // i >= k by construction, rel is monotone
            const double gv = Uc[ok ? ci + (long long)ck * uc : 0];
            g[cc][s2][j] = ok ? gv : 0.0;
          }
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[s2][j] += g[cc][s2][j];
    }
}
__device__ __forceinline__ void pivot_gather_stage2(const PullCtx& pc, const int* invb, int wp, int lane, int wave,
                                                    double (&v)[2][8]) {
  pivot_gather_stage2_from(pc, invb, wp, lane, wave, v, 0);
}
// offsets of the first two children (the others, rare, go the ordinary way behind them)
__device__ __forceinline__ void pivot_offsets_stage2(const PullCtx& pc, const int* invb, int wp, int lane, int wave,
                                                     int (&o2)[2][2][8]) {
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const bool has = cc < pc.n;
    const int uc = pc.uc[cc];
    const int* ib = invb + cc * wp;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int i, k;
        const bool valid = stage2_elem(wp, wave, lane, s2, j, i, k);
        const int ci = (valid && has) ? ib[i] : -1;
        const int ck = (valid && has) ? ib[k] : -1;
        o2[cc][s2][j] = (ci >= 0 && ck >= 0) ? ci + ck * uc : -1;
      }
  }
}

// Dependencies of a workgroup of the single-launch top-of-tree factorisation (k_factor_top): the
// Schur workgroups of the children must have finished before their update matrices are read.
// n == 0 in the per-level kernels (the launch order is the dependency).
__device__ __forceinline__ void flag_wait_ge(int* __restrict__ addr, int target, int* __restrict__ info);
struct ChildWait {
  int n;
  int* addr[MAXCH];
  int target[MAXCH];
  int* info;
  // all children polled by one lane, then ONE acquire (it invalidates L2 lines) and ONE barrier
  __device__ __forceinline__ void wait() const {
    bool any = false;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) any = any || (ch < n && target[ch] > 0);
    if (!any) return;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch)
        if (ch < n && target[ch] > 0) {
          int spins = 0;
          while (__hip_atomic_load(addr[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target[ch]) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) {
              atomicAdd(&info[INFO_TIMEOUT], 1);
              break;
            }
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
};
__device__ __forceinline__ ChildWait no_wait() {
  ChildWait cw;
  cw.n = 0;
  cw.info = nullptr;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) {
    cw.addr[ch] = nullptr;
    cw.target[ch] = 0;
  }
  return cw;
}

// ROWINV (8 waves): the inverse of the unit lower factor is formed block row by block row in
// the shadow of the diagonal-block chain instead of by recursive doubling afterwards.
// ---- LDS flags between the waves of one workgroup (free-running pivot block, V2 below): a wave publishes
// "my tiles are written" with a release + store, a consumer spins on the word.  All lanes execute the same
// scalar-uniform access.  Bounded spins: a hang would take the GPU with it.
// (The data AND the flags live in LDS, whose operations complete in issue order per wave: waiting for the LDS
// counter is all a release needs.  A workgroup-scope fence would also drain the wave's global stores - the tiles it
// has just posted to the panel workgroups, 0.6 us per step of the chain.)
__device__ __forceinline__ void lds_flag_set(int* f, int v) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_add(int* f) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_wait_ge(int* f, int v, int* __restrict__ info) {
  int spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 22)) {
      if ((threadIdx.x & 63) == 0) atomicAdd(&info[INFO_TIMEOUT], 1);
      break;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// LDS words of the free-running pivot block (behind the 1 / d array at Yp)
enum { PF_DIAG = 0, PF_STAGE2 = 1, PF_LRDY = 2, PF_UPD = 10, PF_WORDS = 18 };

__device__ __forceinline__ void dev_pivot_chain(const FrontCtx& c, double* dinv, int* flags, int* __restrict__ info);

template <bool ROWINV, bool CHAIN = false, bool V2 = false>
__device__ __forceinline__ void dev_pivot_block(const FrontCtx& c, int* __restrict__ info, int phases, const PullCtx& pc,
                                                const ChildWait& cw, const PullMore pm = no_more()) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = blockDim.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, r = c.r, wp = c.wp, nbk = c.nbk, lda = c.lda;
  double* A = c.A;
  double* dd = c.dd;
  double* Yp = c.Yp;
  // 32 doubles: the reciprocal pivots of the current diagonal block.  V2 keeps 1 / d of EVERY pivot (the waves
  // run apart): the first wp doubles of the Y panel, which V2 does not use, followed by its flag words
  double* scratch = V2 ? Yp : Yp + 16 * lda;
  int* pflags = reinterpret_cast<int*>(Yp + wp);
  if (V2 && tid < PF_WORDS) pflags[tid] = 0;  // (a barrier follows in every branch below before anybody signals)
  const double* __restrict__ P = c.P;
  if (pc.n > 0) {
    // pull-mode extend-add (8 waves, wp <= 128).  Order of issue = order of the dependent round
    // trips: the children's inverse maps of the pivot rows and the panel entries leave together;
    // the maps go through LDS; then the gathers leave.  Two stages: block column 0 (all threads,
    // 4 entries each) is completed first, so that wave 0 can factor the first diagonal block
    // while waves 1..7 finish the other columns (2 x 16 entries per thread).  Children are added
    // in child order on top of the original entry, as dev_assemble does.
    int* invb = reinterpret_cast<int*>(Yp + 16 * lda + 32);
    int iv[MAXCH];
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) iv[ch] = (ch < pc.n && tid < w) ? pc.inv[ch][tid] : -1;
    const int i1 = tid & 127, kq1 = tid >> 7;  // stage 1: row i1, columns kq1 + 4 q
    double v1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kq1 + 4 * q;
      v1[q] = (i1 == k) ? 1.0 : 0.0;
      if (i1 < w && k < w) v1[q] = (i1 >= k) ? P[i1 + (long long)k * r] : 0.0;
    }
    double v[2][8];  // stage 2 (waves 1..7): lower triangle of the columns 16 .. wp-1, folded (stage2_elem)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int i, k;
        const bool valid = stage2_elem(wp, wave, lane, s2, j, i, k);
        v[s2][j] = (valid && i == k) ? 1.0 : 0.0;
        if (valid && i < w && k < w) v[s2][j] = P[i + (long long)k * r];
      }
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      if (ch < pc.n && tid < wp) invb[ch * wp + tid] = iv[ch];
    __syncthreads();
    if constexpr (!CHAIN) {
      // offsets first, then the wait, then nothing but loads: stage 1 of every child and stage 2 of the first
      // two children leave together, one memory round trip for the lot (the update matrices were written on
      // other XCDs moments ago: a round trip is ~1.5 us, and the index arithmetic of stage 2, ~25 instructions
      // per entry, used to sit between the two)
      int o1[MAXCH][4], o2[2][2][8];
      pivot_offsets_stage1(pc, invb, wp, i1, kq1, o1);
      if (wave >= 1) pivot_offsets_stage2(pc, invb, wp, lane, wave, o2);
      TRW(5);
      cw.wait();  // top-of-tree launch: everything above was requested before the children are awaited
      TRW(1);
      double g1[MAXCH][4], g2[2][2][8];
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch)
        if (ch < pc.n) {
#pragma unroll
          for (int q = 0; q < 4; ++q) g1[ch][q] = pc.Uc[ch][o1[ch][q] >= 0 ? o1[ch][q] : 0];
        }
      if (wave >= 1) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
          if (cc < pc.n) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int j = 0; j < 8; ++j) g2[cc][s2][j] = pc.Uc[cc][o2[cc][s2][j] >= 0 ? o2[cc][s2][j] : 0];
          }
      }
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch)
        if (ch < pc.n) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v1[q] += o1[ch][q] >= 0 ? g1[ch][q] : 0.0;
        }
      if (i1 < wp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) A[i1 + (kq1 + 4 * q) * lda] = v1[q];
      }
      __syncthreads();
      TRP(0);
      if (wave == 0) {
        if (!(phases & 32)) dev_diag_block(c, scratch, 0, info);
        TRP(1);
      } else {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
          if (cc < pc.n) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int j = 0; j < 8; ++j) v[s2][j] += o2[cc][s2][j] >= 0 ? g2[cc][s2][j] : 0.0;
          }
        if (pc.n > 2) pivot_gather_stage2_from(pc, invb, wp, lane, wave, v, 2);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            int i, k;
            if (stage2_elem(wp, wave, lane, s2, j, i, k)) A[i + k * lda] = v[s2][j];
          }
      }
      if (!V2) __syncthreads();  // V2: the chain wave goes on; the others signal PF_STAGE2 below
    } else {
      cw.wait();
      pivot_gather_stage1(pc, invb, wp, i1, kq1, v1);
      // levels with fronts of more than MAXCH children: block after block (child order), everything gathered
      // before anything is stored; the first diagonal block does not overlap the gather here
      if (wave >= 1) pivot_gather_stage2(pc, invb, wp, lane, wave, v);
      for (int nx = pc.next; nx >= 0;) {
        const PullCtx px = make_pull(pm.more[nx], pm.U, pm.inv, pm.rel, 1);
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch)
          if (tid < wp) invb[ch * wp + tid] = (ch < px.n && tid < w) ? px.inv[ch][tid] : -1;
        __syncthreads();
        pivot_gather_stage1(px, invb, wp, i1, kq1, v1);
        if (wave >= 1) pivot_gather_stage2(px, invb, wp, lane, wave, v);
        nx = px.next;
      }
      if (i1 < wp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) A[i1 + (kq1 + 4 * q) * lda] = v1[q];
      }
      if (wave >= 1) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            int i, k;
            if (stage2_elem(wp, wave, lane, s2, j, i, k)) A[i + k * lda] = v[s2][j];
          }
      }
      __syncthreads();
      if (wave == 0 && !(phases & 32)) dev_diag_block(c, scratch, 0, info);
      __syncthreads();
    }
  } else {
    // eight columns per batch so that the panel loads are in flight together
    for (int kk = wave; kk < wp; kk += 8 * nw)
      for (int i = lane; i < wp; i += 64) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = kk + nw * q;
          v[q] = (i == k) ? 1.0 : 0.0;
          if (i < w && k < w) v[q] = (i >= k) ? P[i + (long long)k * r] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = kk + nw * q;
          if (k < wp) A[i + k * lda] = v[q];
        }
      }
    __syncthreads();
    if (wave == 0 && !(phases & 32)) dev_diag_block(c, scratch, 0, info);
    __syncthreads();
  }

  // ROWINV: block row kb of X = inv(L): X[kb, j] = -X_kk sum_{i=j}^{kb-1} L[kb, i] X[i, j], one tile
  // j per wave (waves 1..7), computed during step kb (X_kk is final, wave 0 is busy with the
  // next diagonal block) and written over L[kb, j] at the start of the next step, when nobody
  // reads block row kb of L any more.
  d4_t xpend = {0.0, 0.0, 0.0, 0.0};
  int xrow = -1, xcol = -1;
  if constexpr (V2) {
    if (wave >= 1) lds_flag_add(&pflags[PF_STAGE2]);  // this wave's share of the block is in LDS
    dev_pivot_chain(c, scratch, pflags, info);
    TRW(4);
    __syncthreads();
  }
  for (int kb = 0; kb < (V2 ? 0 : nbk); ++kb) {
    const int k0 = kb << 4;
    if (ROWINV && xrow >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * xrow + lk + 4 * q, col = 16 * xcol + li;
        A[row + col * lda] = -xpend[q];
        if (row < w) {
          c.P[row + (long long)col * r] = -xpend[q];  // final: straight to the panel as well
          if (c.Xa) post_f64(c.Xa + row + col * c.wp, -xpend[q]);  // ... and to the panel workgroups that poll for it
        }
      }
      xrow = -1;
    }
    // S2: block column.  Y_Ik = A_Ik X_kk^T, L_Ik = Y_Ik D^-1.
    if (!(phases & 64))
      for (int I = kb + 1 + wave; I < nbk; I += nw) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double av = A[(16 * I + li) + (k0 + 4 * s + lk) * lda];
          const double bv = A[(k0 + li) + (k0 + 4 * s + lk) * lda];
          acc = MFMA_F64(av, bv, acc);
        }
        const double dinv = scratch[li];  // left by the diagonal block
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * I + lk + 4 * q;
          Yp[row + li * lda] = acc[q];
          A[row + (k0 + li) * lda] = acc[q] * dinv;
        }
      }
    __syncthreads();
    // S3 + look-ahead S1: tile t = 0 is the next diagonal block (kb+1, kb+1)
    if (kb + 1 < nbk) {
      const int T = nbk - kb - 1;
      const int ntiles = T * (T + 1) / 2;
      if (wave == 0) {
        if (!(phases & 64)) dev_trailing_tile(c, k0, kb + 1, kb + 1);
        if (!(phases & 32)) dev_diag_block(c, scratch, k0 + 16, info);
      }
      if (!(phases & 64)) {
        // remaining tiles over the other waves (over all waves when there is only one)
        // wave 0 carries the sequential chain; with eight waves, wave 4 shares its SIMD (and the
        // fp64 pipe that the MFMAs of a trailing tile keep busy), so it takes no tiles either
        const bool quiet4 = (nw == 8);
        const int widx = quiet4 ? (wave < 4 ? wave - 1 : wave - 2) : wave - 1;
        const int first = (nw > 1) ? 1 + widx : 1;
        const int step = (nw > 1) ? (quiet4 ? 6 : nw - 1) : 1;
        if (nw == 1 || (wave > 0 && !(quiet4 && wave == 4)))
          for (int t = first; t < ntiles; t += step) {
            int J = 0, rem = t;
            while (rem >= T - J) {
              rem -= T - J;
              ++J;
            }
            dev_trailing_tile(c, k0, kb + 1 + J + rem, kb + 1 + J);
          }
      }
    }
    if (ROWINV && wave == 1 + (kb + 3) % 7) {
      // diagonal block kb is final since S1(kb): inverse of its unit factor below, pivots on the diagonal
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = k0 + li, col = k0 + lk + 4 * q;
        if (row < w && col <= row) {
          const double val = (row == col) ? dd[col] : A[row + col * lda];
          c.P[row + (long long)col * r] = val;
          if (c.Xa) post_f64(c.Xa + row + col * c.wp, val);
        }
      }
    }
    // tile j of the row on wave j + 1, except that wave 4 stays off the SIMD of the chain: it takes
    // the tile that only exists in the last step (j = 6, kb = 7), when no diagonal block is in flight
    const int xj = (wave < 4) ? wave - 1 : (wave == 4 ? 6 : wave - 2);
    if (ROWINV && wave >= 1 && xj < kb && !(phases & 128)) {
      const int j = xj;
      d4_t t = {0.0, 0.0, 0.0, 0.0};
      for (int i = j; i < kb; ++i)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          const double av = A[(k0 + li) + (16 * i + 4 * s2 + lk) * lda];      // L[kb, i]
          const double bv = A[(16 * i + 4 * s2 + lk) + (16 * j + li) * lda];  // X[i, j] (X_jj for i = j)
          t = MFMA_F64(av, bv, t);
        }
      d4_t x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) x = MFMA_F64(A[(k0 + li) + (k0 + 4 * s2 + lk) * lda], t[s2], x);  // X_kk T
      xpend = x;
      xrow = kb;
      xcol = j;
    }
    __syncthreads();
  }
  if (ROWINV && xrow >= 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * xrow + lk + 4 * q, col = 16 * xcol + li;
      A[row + col * lda] = -xpend[q];
      if (row < w) {
        c.P[row + (long long)col * r] = -xpend[q];
        if (c.Xa) post_f64(c.Xa + row + col * c.wp, -xpend[q]);
      }
    }
  }
  // Pivot range of the front (condition estimate, refinement tolerance): behind the info words,
  // word 0 = max over ~bits(|d|) (the minimum), word 1 = max over bits(|d|), both zeroed with them.
  // Bit patterns of non-negative doubles order like the values, so integer atomics do, in any
  // order of arrival.  One wave that is off the chain, two atomics without return value.
  if (wave == nw - 1 && !(phases & 32)) {
    double lo = 1.7e308, hi = 0.0;
    for (int k = lane; k < w; k += 64) {
      const double d = fabs(dd[k]);
      lo = fmin(lo, d);
      hi = fmax(hi, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = fmin(lo, __shfl_down(lo, o, 64));
      hi = fmax(hi, __shfl_down(hi, o, 64));
    }
    if (lane == 0) {
      unsigned long long* mm = reinterpret_cast<unsigned long long*>(info + INFO_WORDS);
      atomicMax(&mm[0], ~(unsigned long long)__double_as_longlong(lo));
      atomicMax(&mm[1], (unsigned long long)__double_as_longlong(hi));
    }
  }
  if (ROWINV) __syncthreads();

  // inverse of the unit lower block factor by recursive doubling.  A holds
  // inv(L_kk) in the diagonal blocks and L_IJ below.  For [X11 0; B X22] the
  // off-diagonal block of the inverse is -X22 B X11; one wave computes one
  // 16-column strip of it in registers (the accumulator tiles of the first
  // product are the B operands of the second), then all strips are stored.
  for (int h = 16; h < wp && !(phases & 128) && !ROWINV; h <<= 1) {
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
}


// ---- V2 of the blocked LDL^T of the pivot block: free-running waves instead of two workgroup barriers per
// block column.  The dependent chain of a front is diag(kb) -> block (kb+1, kb) of L -> update of the diagonal
// tile (kb+1, kb+1) -> diag(kb+1); with barriers the chain wave also waited for everybody else's share of every
// step (2.4 - 2.6 us per block column, of which the chain itself needs ~1.5).  Here
//   wave 0 owns the chain: after diag(kb) it forms block (kb+1, kb) of L itself - transposed, Y^T = X_kk A^T, so
//          that the accumulator registers are at once the A operand (L) and the B operand (Y) of the update of
//          the diagonal tile, which never travels through LDS in between -, then diag(kb+1);
//   wave I (1..7) owns block row I of the trailing matrix for good: in step kb < I-1 it forms L(I, kb) and
//          updates its tiles (I, kb+1 .. I); nobody else ever writes that row, so no barrier is needed, only
//          "tile (J, kb) of L is there" (PF_LRDY) for the B operands of the other rows J < I (= D L_J^T, formed
//          from the L tile: tiles of L are written once and never overwritten, a Y panel would be reused);
//   row kb+1 is handed to wave 0 once its owner has applied the steps < kb (PF_UPD).
// Leaves L (unit lower, off-diagonal tiles), inv(L_kk) in the diagonal tiles, the pivots in dd and 1 / d in
// dinv; the caller inverts L afterwards (recursive doubling, off the critical path: the panel workgroups of
// the dataflow launch follow the POSTED tiles of L and solve by block substitution, dev_panel_rows_subst).
// Needs 8 waves (one per block row, wp <= 128).
__device__ __forceinline__ void dev_pivot_post_diag(const FrontCtx& c, int k0) {
  if (!c.Xa) return;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const int row = k0 + li;
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = c.A[row + (k0 + lk + 4 * q) * c.lda];  // (one batch of LDS reads, then the stores)
  const double dv = c.dd[row];
  if (row >= c.w) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = k0 + lk + 4 * q;
    if (col < row) post_f64(c.Xa + row + col * c.wp, v[q]);
  }
  if (lk == 0) post_f64(c.Xa + row + row * c.wp, dv);
}

__device__ __forceinline__ void dev_pivot_chain(const FrontCtx& c, double* dinv, int* flags, int* __restrict__ info) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, wp = c.wp, nbk = c.nbk, lda = c.lda;
  double* A = c.A;
  if (wave == 0) {
    // (the inverse of the diagonal block stays in the registers: it is the A operand of the next block of L)
    double x[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) x[cc] = A[li + (4 * cc + lk) * lda];
    lds_flag_set(&flags[PF_DIAG], 1);
    for (int kb = 0; kb + 1 < nbk; ++kb) {
      const int k0 = kb << 4, r0 = k0 + 16;
      if (kb == 0)
        lds_flag_wait_ge(&flags[PF_STAGE2], 7, info);
      else
        lds_flag_wait_ge(&flags[PF_UPD + kb + 1], kb, info);
      TRP(2 + 3 * kb);
      // Y^T = X_kk A(kb+1, kb)^T: y[q] = Y[li][lk + 4 q]; the diagonal tile it updates is requested with the operands
      double bv[4], di[4];
      d4_t t;
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = A[(r0 + li) + (k0 + 4 * s + lk) * lda];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        di[q] = dinv[k0 + lk + 4 * q];
        t[q] = A[(r0 + lk + 4 * q) + (r0 + li) * lda];
      }
      d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) y = MFMA_F64(x[s], bv[s], y);
      d4_t l;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        l[q] = y[q] * di[q];
        A[(r0 + li) + (k0 + lk + 4 * q) * lda] = l[q];
      }
      lds_flag_set(&flags[PF_LRDY + kb + 1], kb + 1);
      // diagonal tile (kb+1, kb+1) -= L Y^T, operands straight from the registers (k runs as lk + 4 s on both)
#pragma unroll
      for (int s = 0; s < 4; ++s) t = MFMA_F64(l[s], -y[s], t);
#pragma unroll
      for (int q = 0; q < 4; ++q) A[(r0 + lk + 4 * q) + (r0 + li) * lda] = t[q];
      if (c.Xa && r0 + li < w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) post_f64(c.Xa + (r0 + li) + (k0 + lk + 4 * q) * wp, l[q]);
      }
      TRP(3 + 3 * kb);
      dev_diag_block(c, dinv + r0, r0, info, x);
      TRP(4 + 3 * kb);
      lds_flag_set(&flags[PF_DIAG], kb + 2);
    }
  } else if (wave != 4 && (wave < 4 ? wave + 1 : wave) < nbk) {
    // rows 2 .. 7 on the waves 1, 2, 3, 5, 6, 7 (row 1 goes straight to the chain wave).  Wave 4 shares the SIMD of
    // the chain wave - its matrix instructions would stretch the chain, and the chain's steady issue starves it until
    // its row is the one the chain waits for - and posts the diagonal tiles instead.
    const int I = wave < 4 ? wave + 1 : wave, i0 = I << 4;
    for (int kb = 0; kb + 1 < I; ++kb) {
      const int k0 = kb << 4;
      lds_flag_wait_ge(&flags[PF_DIAG], kb + 1, info);
      d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double av = A[(k0 + li) + (k0 + 4 * s + lk) * lda];
        const double bv = A[(i0 + li) + (k0 + 4 * s + lk) * lda];
        y = MFMA_F64(av, bv, y);
      }
      d4_t l;
      double dk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        l[q] = y[q] * dinv[k0 + lk + 4 * q];
        dk[q] = c.dd[k0 + lk + 4 * q];
        A[(i0 + li) + (k0 + lk + 4 * q) * lda] = l[q];
      }
      lds_flag_set(&flags[PF_LRDY + I], kb + 1);
      if (c.Xa && i0 + li < w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) post_f64(c.Xa + (i0 + li) + (k0 + lk + 4 * q) * wp, l[q]);
      }
      if (kb == 0) lds_flag_wait_ge(&flags[PF_STAGE2], 7, info);  // the tiles right of block column 0 were loaded by everybody
      // own diagonal tile first (needs nothing from the other rows), then the tiles J = kb+1 .. I-1
      for (int jj = 0; jj <= I - kb - 1; ++jj) {
        const int J = (jj == 0) ? I : kb + jj;
        const int j0 = J << 4;
        d4_t t;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = A[(i0 + lk + 4 * q) + (j0 + li) * lda];
        if (J < I) {
          lds_flag_wait_ge(&flags[PF_LRDY + J], kb + 1, info);
#pragma unroll
          for (int s = 0; s < 4; ++s) t = MFMA_F64(l[s], -(A[(j0 + li) + (k0 + lk + 4 * s) * lda] * dk[s]), t);
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) t = MFMA_F64(l[s], -y[s], t);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) A[(i0 + lk + 4 * q) + (j0 + li) * lda] = t[q];
      }
      lds_flag_set(&flags[PF_UPD + I], kb + 1);
#ifdef HIPFACT_TRACE
      if (lane == 0 && kb + 2 == I && blockIdx.x < TRACE_WGS) g_own[blockIdx.x * 8 + I] = wall_clock64();
#endif
    }
  }
  // The inverses of the diagonal blocks and the pivots are posted to the panel workgroups by wave 4: five LDS reads
  // and stores per block that the chain wave does not have to issue.
  if (c.Xa && wave == 4) {
    for (int kb = 0; kb < nbk; ++kb) {
      lds_flag_wait_ge(&flags[PF_DIAG], kb + 1, info);
      dev_pivot_post_diag(c, kb << 4);
    }
  }
}

// store inv(L11) (strict lower) and the pivots (diagonal) back to the panel
__device__ __forceinline__ void dev_store_pivot_block(const FrontCtx& c) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  for (int k = wave; k < c.w; k += nw)
    for (int i = k + lane; i < c.w; i += 64)
      c.P[i + (long long)k * c.r] = (i == k) ? c.dd[k] : c.A[i + k * c.lda];
}

// reload inv(L11) and the pivots from a finished panel (split kernels)
// recip: dd receives 1 / d_k (the split panel kernels scale by multiplication: one division per
// pivot and workgroup instead of one per entry of L21)
__device__ __forceinline__ void dev_load_pivot_block(const FrontCtx& c, bool need_x, bool recip = false) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = blockDim.x >> 6;
  for (int k = tid; k < c.wp; k += blockDim.x) {
    const double d = (k < c.w) ? c.P[k + (long long)k * c.r] : 1.0;
    c.dd[k] = recip ? 1.0 / d : d;
  }
  if (need_x && nw == 8) {
    // 8 waves, wp <= 128: the whole block as ONE batch of 2 x 16 loads per thread (each pass of the
    // generic loop below is a dependent memory round trip on the critical path of the panel solve)
    double v[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = lane + 64 * t, k = wave + 8 * q;
        v[t][q] = (i == k) ? 1.0 : 0.0;
        if (i < c.w && k < c.w && i > k) v[t][q] = c.P[i + (long long)k * c.r];
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = lane + 64 * t, k = wave + 8 * q;
        if (i < c.wp && k < c.wp) c.A[i + k * c.lda] = v[t][q];
      }
  } else if (need_x)
    for (int kk = wave; kk < c.wp; kk += 8 * nw)
      for (int i = lane; i < c.wp; i += 64) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = kk + nw * q;
          v[q] = (i == k) ? 1.0 : 0.0;
          if (i < c.w && k < c.w && i > k) v[q] = c.P[i + (long long)k * c.r];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = kk + nw * q;
          if (k < c.wp) c.A[i + k * c.lda] = v[q];
        }
      }
  __syncthreads();
}

// rows R0 + li of the panel: all fragments in one batch of loads (w <= 128: 32 values per lane),
// plus the children's rows that land on them (pull mode)
// DL: element (tt, s) of a lane is column 16 tt + lk + 4 s (the accumulator layout of the matrix cores, rows of the
// transposed block) instead of 16 tt + 4 s + lk (the B operand layout)
template <bool DL = false>
__device__ __forceinline__ void dev_panel_rows_load(const FrontCtx& c, int R0, const PullCtx& pc, double (&pv)[8][4],
                                                    int (&cis)[MAXCH]) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, r = c.r;
  const bool rok = (R0 + li) < r;
  const double* __restrict__ Prow = c.P + R0 + li;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) cis[ch] = (ch < pc.n && rok) ? pc.inv[ch][R0 + li] : -1;
#pragma unroll
  for (int tt = 0; tt < 8; ++tt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int col = 16 * tt + (DL ? lk + 4 * s : 4 * s + lk);
      pv[tt][s] = (rok && col < w) ? Prow[(long long)col * r] : 0.0;
    }
}

// gathers (pull mode), X P21^T on the MFMA units, scaling by D^-1, row-contiguous stores
// the children's rows that land on the panel rows R0 + li (pull mode), added in child order
template <bool DL = false>
__device__ __forceinline__ void dev_panel_rows_gather(const FrontCtx& c, const PullCtx& pc, const int* invl,
                                                      double (&pv)[8][4], const int (&cis)[MAXCH]) {
  const int lane = threadIdx.x & 63;
  const int lk = lane >> 4;
  const int nbk = c.nbk;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (ch < pc.n) {
      // unconditional loads (clamped to entry 0) so that they all leave in one batch
      const int ci = cis[ch];
      const double* __restrict__ Uc = pc.Uc[ch];
      const int uc = pc.uc[ch];
      double g[8][4];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int col = 16 * tt + (DL ? lk + 4 * s : 4 * s + lk);
          const int cj = (tt < nbk && ci >= 0) ? invl[ch * c.wp + col] : -1;
          const double gv = Uc[(cj >= 0) ? ci + (long long)cj * uc : 0];
          g[tt][s] = (cj >= 0) ? gv : 0.0;
        }
#pragma unroll
      for (int tt = 0; tt < 8; ++tt)
#pragma unroll
        for (int s = 0; s < 4; ++s) pv[tt][s] += g[tt][s];
    }
}

// X P21^T on the MFMA units, scaling by D^-1, row-contiguous stores
// cstep / c0: this wave computes the 16-column output blocks ct == c0 (mod cstep) only (two waves
// share a strip of rows in the top-of-tree launch, where workgroups are plentiful and the MFMA time
// of a strip sits on the critical path)
template <bool X_IN_LDS, bool RECIP = false>
__device__ __forceinline__ void dev_panel_rows_product(const FrontCtx& c, int R0, double (&pv)[8][4], int cstep = 1,
                                                       int c0 = 0) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, r = c.r, nbk = c.nbk, lda = c.lda;
  const double* A = c.A;
  double* __restrict__ P = c.P;
  const bool rok = (R0 + li) < r;
  d4_t acc[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) acc[ct] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int tt = 0; tt < 8; ++tt)
    if (tt < nbk) {
#pragma unroll
      for (int ct = 0; ct < 8; ++ct)
        if (ct >= tt && ct < nbk) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double xv = A[(16 * ct + li) + (16 * tt + 4 * s + lk) * lda];
            acc[ct] = MFMA_F64(xv, pv[tt][s], acc[ct]);
          }
        }
    }
  if (rok) {
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
      if (ct < nbk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 16 * ct + lk + 4 * q;
          if (col < w) {
            const double dcol = X_IN_LDS ? c.dd[col] : P[col + (long long)col * r];
            P[(R0 + li) + (long long)col * r] = RECIP ? acc[ct][q] * dcol : acc[ct][q] / dcol;  // RECIP: dd holds 1 / d
          }
        }
      }
  }
}

// Panel rows in the single-launch factorisation.  The pivot workgroup posts inv(L11) tile by tile
// as the tiles become final (block row ct after step ct of its loop), and output block ct of
// L21 = P21 inv(L11)^T D^-1 needs exactly block row ct: the panel workgroups poll the tiles
// themselves (no flag, no fence, see poll_f64) and follow the pivot workgroup block row by block
// row, so that only the last block row is left when the pivot block is finished.  Same products
// in the same order as dev_panel_rows_product<true, true> (ct-major instead of tt-major nesting:
// every accumulator still sees tt ascending).
__device__ __forceinline__ void dev_panel_rows_product_posted(const FrontCtx& c, int R0, double (&pv)[8][4], int cstep,
                                                              int c0, bool active, int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, r = c.r, nbk = c.nbk, lda = c.lda, wp = c.wp;
  double* A = c.A;
  double* __restrict__ P = c.P;
  const bool rok = active && (R0 + li) < r;
  for (int ct = 0; ct < nbk; ++ct) {
    // block row ct of X = inv(L11): tiles (ct, 0 .. ct), unit diagonal, and 1 / d of its 16 pivots
    for (int e = tid; e < (ct + 1) * 256; e += blockDim.x) {
      const int tt = e >> 8, i = e & 15, k = (e >> 4) & 15;
      const int row = 16 * ct + i, col = 16 * tt + k;
      double v = (row == col) ? 1.0 : 0.0;
      if (row < w && col < row) v = poll_f64(c.Xa + row + col * wp, info);
      A[row + col * lda] = v;
    }
    if (tid < 16) {
      const int col = 16 * ct + tid;
      c.dd[col] = (col < w) ? 1.0 / poll_f64(c.Xa + col + col * wp, info) : 1.0;
    }
    __syncthreads();
    if (active && (ct % cstep) == c0) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int tt = 0; tt < 8; ++tt)
        if (tt <= ct) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double xv = A[(16 * ct + li) + (16 * tt + 4 * s + lk) * lda];
            acc = MFMA_F64(xv, pv[tt][s], acc);
          }
        }
      if (rok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 16 * ct + lk + 4 * q;
          if (col < w) P[(R0 + li) + (long long)col * r] = acc[q] * c.dd[col];
        }
      }
    }
  }
}


// Panel rows in the single-launch factorisation, V2: the pivot workgroup posts the tiles of L11 itself (unit lower
// factor: tile (I, kb) right after step kb, the inverse of the diagonal block kb and its pivots after diag(kb)) and no
// longer forms inv(L11) on the critical path.  The panel is solved by block substitution, transposed so that a wave's
// finished blocks are at once the B operands of the later ones: with W = L21 D,  L11 W^T = P21^T, i.e.
//   W^T[ct] = X_ct (P21^T[ct] - sum_{tt < ct} L11[ct, tt] W^T[tt]),   X_ct = inv(L11[ct, ct]),
// every accumulator (rows lk + 4 q of the block = columns of L21, column li = panel row) serves as operand with k
// running as lk + 4 s.  Block row ct of L11 is complete when diag(ct) has been posted, so the panel follows the pivot
// workgroup block by block and 8 dependent matrix instructions remain when the pivot block is done.  pv: the wave's
// 16 panel rows in the accumulator layout (dev_panel_rows_load<true>), overwritten by W^T.
// store_d: this workgroup also writes the pivots it polled onto the diagonal of the panel.  The Schur workgroups read
// d_k from there once the PANEL workgroups have published, and the pivot workgroup - busy inverting L11 behind its
// chain - stores its copy (the same bits) only later.
__device__ __forceinline__ void dev_panel_rows_subst_posted(const FrontCtx& c, int R0, double (&pv)[8][4], bool active,
                                                            bool store_d, int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, r = c.r, nbk = c.nbk, lda = c.lda, wp = c.wp;
  double* A = c.A;
  double* __restrict__ P = c.P;
  const bool rok = active && (R0 + li) < r;
  // Polling: thread t owns position (pi, pk) of the tiles tq, tq + 2, tq + 4, tq + 6 of every block row, four requests
  // per block row that leave together (a poll per element would be a memory round trip each); what still holds the
  // sentinel is requested again, pass after pass.  The requests for block row ct + 1 are issued before the matrix work
  // of row ct and stay in flight across it.  (Kept to a few dozen instructions per row: with eight waves on four
  // SIMDs a wave issues one instruction per ~8 clocks, and a general element -> thread map with per-element
  // predicates cost more than the round trips it saved.)
  constexpr int NB = 4;
  constexpr unsigned long long ONE = 0x3FF0000000000000ull;
  const int pi = tid & 15, pk = (tid >> 4) & 15, tq = tid >> 8;
  const unsigned long long* xbase = reinterpret_cast<const unsigned long long*>(c.Xa) + pi + pk * wp;
  const unsigned long long* xdiag = reinterpret_cast<const unsigned long long*>(c.Xa) + tid * (wp + 1);
  unsigned long long nb[8][NB], nd[8];
  auto addr = [&](int cr, int b2) { return xbase + 16 * cr + 16 * (tq + 2 * b2) * wp; };
  auto need = [&](int cr, int b2) {
    const int tt = tq + 2 * b2;
    return 16 * cr + pi < w && (tt < cr || (tt == cr && pk < pi));
  };
  // A panel workgroup gets going ~8 us after the pivot workgroup (its own rows and the children's contributions are four
  // dependent round trips): the block rows posted by then are all requested at once, before the first one is used
#pragma unroll
  for (int cr = 0; cr < 8; ++cr) {
    if (cr >= nbk) break;
#pragma unroll
    for (int b2 = 0; b2 < NB; ++b2) {
      const int tt = tq + 2 * b2;
      nb[cr][b2] = need(cr, b2) ? __hip_atomic_load(addr(cr, b2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : ((tt == cr && pk == pi) ? ONE : 0ull);  // unit diagonal, zero padding
    }
    nd[cr] = (tid < 16 && 16 * cr + tid < w) ? __hip_atomic_load(xdiag + 16 * cr * (wp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ONE;
  }
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) {  // (unrolled: pv, nb, nd are indexed by ct and must stay in registers)
    if (ct >= nbk) break;
    // block row ct of L11: tiles (ct, 0 .. ct-1), the inverse of the diagonal block (unit diagonal), 1 / d of its pivots
    for (int spins = 0;; ++spins) {
      bool again = false;
#pragma unroll
      for (int b2 = 0; b2 < NB; ++b2)
        if (nb[ct][b2] == SOLVE_SENT) {
          nb[ct][b2] = __hip_atomic_load(addr(ct, b2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          again = true;
        }
      if (nd[ct] == SOLVE_SENT) {
        nd[ct] = __hip_atomic_load(xdiag + 16 * ct * (wp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        again = true;
      }
      if (!again) break;
      if (spins > (1 << 20)) {
        atomicAdd(&info[INFO_TIMEOUT], 1);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int b2 = 0; b2 < NB; ++b2) {
      const int tt = tq + 2 * b2;
      if (tt <= ct) A[(16 * ct + pi) + (16 * tt + pk) * lda] = __longlong_as_double((long long)nb[ct][b2]);
    }
    if (tid < 16) {
      const int col = 16 * ct + tid;
      const double d = __longlong_as_double((long long)nd[ct]);
      c.dd[col] = 1.0 / d;
      if (store_d && col < w) P[col + (long long)col * r] = d;
    }
    TRP(8 + ct);
    __syncthreads();
    TRP(ct);
    // what was not there yet of the next block row is requested again; in flight across the matrix work of this one
    if (ct + 1 < nbk) {
#pragma unroll
      for (int b2 = 0; b2 < NB; ++b2)
        if (nb[ct + 1 < 8 ? ct + 1 : 7][b2] == SOLVE_SENT)
          nb[ct + 1 < 8 ? ct + 1 : 7][b2] = __hip_atomic_load(addr(ct + 1, b2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nd[ct + 1 < 8 ? ct + 1 : 7] == SOLVE_SENT)
        nd[ct + 1 < 8 ? ct + 1 : 7] = __hip_atomic_load(xdiag + 16 * (ct + 1) * (wp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (active) {
      // operands first (one batch of LDS reads), then the matrix instructions on four accumulators, one per k
      // group: a dependent matrix instruction waits for its predecessor's result, and so does one behind a load
      double lv[8][4], xv[4];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt)
        if (tt < ct) {
#pragma unroll
          for (int s = 0; s < 4; ++s) lv[tt][s] = -A[(16 * ct + li) + (16 * tt + lk + 4 * s) * lda];
        }
#pragma unroll
      for (int s = 0; s < 4; ++s) xv[s] = A[(16 * ct + li) + (16 * ct + lk + 4 * s) * lda];
      d4_t acc = {pv[ct][0], pv[ct][1], pv[ct][2], pv[ct][3]};
      d4_t ac[3] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
      for (int tt = 0; tt < 8; ++tt)
        if (tt < ct) {
          acc = MFMA_F64(lv[tt][0], pv[tt][0], acc);
#pragma unroll
          for (int s = 1; s < 4; ++s) ac[s - 1] = MFMA_F64(lv[tt][s], pv[tt][s], ac[s - 1]);
        }
      if (ct > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += (ac[0][q] + ac[1][q]) + ac[2][q];
      }
      d4_t res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) res = MFMA_F64(xv[s], acc[s], res);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pv[ct][q] = res[q];
        const int col = 16 * ct + lk + 4 * q;
        if (rok && col < w) P[(R0 + li) + (long long)col * r] = res[q] * c.dd[col];
      }
    }
    TRP(16 + ct);
  }
}

template <bool X_IN_LDS, bool RECIP = false>
__device__ __forceinline__ void dev_panel_rows_finish(const FrontCtx& c, int R0, const PullCtx& pc, const int* invl,
                                                      double (&pv)[8][4], const int (&cis)[MAXCH]) {
  dev_panel_rows_gather(c, pc, invl, pv, cis);
  dev_panel_rows_product<X_IN_LDS, RECIP>(c, R0, pv);
}

// ---- phase C: L21^T tiles = X P21^T, scaled by D^-1.  Each wave owns 16 panel
// rows: the B operand streams from the panel (16 consecutive rows per k-step),
// the A operand is X = inv(L11) in LDS.  Row blocks blk, blk + stride, ...
template <bool X_IN_LDS, bool RECIP = false>
__device__ __forceinline__ void dev_panel_solve(const FrontCtx& c, int blk, int blk_stride, const PullCtx& pc,
                                                const int* invl) {
  const int wave = threadIdx.x >> 6;
  const int RB = 16 * (blockDim.x >> 6);  // panel rows per workgroup pass
  for (int R0 = c.w + RB * blk + 16 * wave; R0 < c.r; R0 += RB * blk_stride) {
    double pv[8][4];
    int cis[MAXCH];
    dev_panel_rows_load(c, R0, pc, pv, cis);
    dev_panel_rows_finish<X_IN_LDS, RECIP>(c, R0, pc, invl, pv, cis);
  }
}

// ---- phase D: one 64 x 64 tile (I, J) of U_s -= L21 D L21^T.  Operand strips
// (64 rows x 64 pivots per chunk, k-major) staged in LDS: 64 KB per workgroup, so
// that two workgroups share a CU and hide each other's staging latency.
constexpr int KC = 32;
// children's contributions to the lane's 2 x 2 x 4 tile entries, added in child order; invs =
// LDS copy of the children's inverse maps of the tile's rows ([0, 64)) and columns ([64, 128))
__device__ __forceinline__ void schur_tile_gather(const PullCtx& pc, const int* invs, int i0, int j0, int li, int lk,
                                                  double (&uv)[2][2][4]) {
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (ch < pc.n) {
      const double* __restrict__ Uc = pc.Uc[ch];
      const int uc = pc.uc[ch];
      const int* iv = invs + 128 * ch;
      int ci[2], cj[2][4];
#pragma unroll
      for (int y = 0; y < 2; ++y) ci[y] = iv[i0 + 16 * y + li];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int q = 0; q < 4; ++q) cj[x][q] = iv[64 + j0 + 16 * x + lk + 4 * q];
      double g[2][2][4];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // rel is monotone: i >= j in the front implies ci >= cj in the child; loads are
            // unconditional (clamped to entry 0) so that they all leave in one batch
            const bool ok = ci[y] >= 0 && cj[x][q] >= 0 && ci[y] >= cj[x][q];
            const double gv = Uc[ok ? ci[y] + (long long)cj[x][q] * uc : 0];
            g[x][y][q] = ok ? gv : 0.0;
          }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int q = 0; q < 4; ++q) uv[x][y][q] += g[x][y][q];
    }
}

template <bool DD_IN_LDS, bool CHAIN = false>
__device__ __forceinline__ void dev_schur_tile(const FrontCtx& c, double* SI, double* SJ, int I, int J, bool assign,
                                               const PullCtx& pc, const int tid, int* wait_addr = nullptr,
                                               int wait_target = 0, int* info = nullptr,
                                               const PullMore pm = no_more()) {
  // wait_addr (single-launch top-of-tree factorisation): the children's entries are gathered
  // first, then the panel workgroups of the own front are awaited, then the operands are staged
  // tid: thread index inside the 256-thread team that owns the tile (barriers stay workgroup-wide)
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, r = c.r, u = c.u;
  const double* __restrict__ P21 = c.P + w;
  const int wi = wave & 1, wj = wave >> 1;
  const bool ghost = I < 0;  // team without a tile: takes part in the barriers only
  const bool idle = ghost || (I == J && wi < wj);  // block above the diagonal
  const int i0 = 32 * wi, j0 = 32 * wj;
  d4_t acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = (d4_t){0.0, 0.0, 0.0, 0.0};
  // the assembled U tile is fetched up front so that its latency hides behind the operand staging
  double uv[2][2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int i = 64 * I + i0 + 16 * y + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 64 * J + j0 + 16 * x + lk + 4 * q;
        uv[x][y][q] = (!assign && !idle && pc.n == 0 && i < u && j < u && i >= j) ? c.Us[i + (long long)j * u] : 0.0;
      }
    }
  // pull mode: the children's inverse maps of the tile's 64 rows and 64 columns go through LDS
  // (two entries per thread); the gathers themselves run after the MFMA work, so that the operand
  // staging is not delayed by their two dependent round trips
  int* invs = reinterpret_cast<int*>(SJ + 64 * KC);  // [child][0..63: rows | 64..127: columns]
  int ivr[2] = {-1, -1};
  if (pc.n > 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch = (tid >> 7) + 2 * h, idx = tid & 127;
      const int gi = (idx < 64) ? 64 * I + idx : 64 * J + idx - 64;
#pragma unroll
      for (int cc = 0; cc < MAXCH; ++cc)
        if (cc == ch && cc < pc.n && gi < u && !ghost) ivr[h] = pc.inv[cc][w + gi];
    }
  }
  if (wait_addr) {
    if (pc.n > 0) {
      invs[tid] = ivr[0];
      invs[tid + 256] = ivr[1];
    }
    __syncthreads();
    if (!idle) schur_tile_gather(pc, invs, i0, j0, li, lk, uv);
    TRW(5);
    flag_wait_ge(wait_addr, wait_target, info);
    TRW(1);
  }
  const int si = tid & 63;
  const bool iok = !ghost && (64 * I + si) < u, jok = !ghost && (64 * J + si) < u;
  const double* __restrict__ pi = P21 + 64 * I + si;
  const double* __restrict__ pj = P21 + 64 * J + si;
  for (int kc0 = 0; kc0 < w; kc0 += KC) {
    const int kcn = min(KC, w - kc0);
    __syncthreads();  // previous chunk / tile has finished reading the strips
    for (int kk = tid >> 6; kk < kcn; kk += 32) {
      double vi[8], vj[8], vd[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = kk + 4 * q;
        vi[q] = (iok && k < kcn) ? pi[(long long)(kc0 + k) * r] : 0.0;
        vj[q] = (jok && k < kcn) ? pj[(long long)(kc0 + k) * r] : 0.0;
        // pivot d_k: LDS copy (fused kernel) or the panel diagonal (wave-uniform address)
        vd[q] = (k < kcn) ? (DD_IN_LDS ? c.dd[kc0 + k] : c.P[(kc0 + k) + (long long)(kc0 + k) * r]) : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = kk + 4 * q;
        if (k < kcn) {
          SI[k * 64 + si] = vi[q];
          SJ[k * 64 + si] = vj[q] * vd[q];
        }
      }
    }
    if (kc0 == 0 && pc.n > 0 && !wait_addr) {
      invs[tid] = ivr[0];
      invs[tid + 256] = ivr[1];
    }
    __syncthreads();
    if (!idle) {
      const int k4 = kcn & ~3;
      for (int k0 = 0; k0 < k4; k0 += 4) {
        const int kk = (k0 + lk) * 64;
        const double a0 = SJ[kk + j0 + li], a1 = SJ[kk + j0 + 16 + li];
        const double b0 = SI[kk + i0 + li], b1 = SI[kk + i0 + 16 + li];
        acc[0][0] = MFMA_F64(a0, b0, acc[0][0]);
        acc[0][1] = MFMA_F64(a0, b1, acc[0][1]);
        acc[1][0] = MFMA_F64(a1, b0, acc[1][0]);
        acc[1][1] = MFMA_F64(a1, b1, acc[1][1]);
      }
      if (k4 < kcn) {
        const bool kok = (k4 + lk) < kcn;
        const int kk = (k4 + lk) * 64;
        const double a0 = kok ? SJ[kk + j0 + li] : 0.0, a1 = kok ? SJ[kk + j0 + 16 + li] : 0.0;
        const double b0 = kok ? SI[kk + i0 + li] : 0.0, b1 = kok ? SI[kk + i0 + 16 + li] : 0.0;
        acc[0][0] = MFMA_F64(a0, b0, acc[0][0]);
        acc[0][1] = MFMA_F64(a0, b1, acc[0][1]);
        acc[1][0] = MFMA_F64(a1, b0, acc[1][0]);
        acc[1][1] = MFMA_F64(a1, b1, acc[1][1]);
      }
    }
  }
  if (!wait_addr) {
    if (!idle) schur_tile_gather(pc, invs, i0, j0, li, lk, uv);
    // fronts with more than MAXCH children: further blocks of children, in child order
    for (int nx = CHAIN ? pc.next : -1; nx >= 0;) {
      const PullCtx px = make_pull(pm.more[nx], pm.U, pm.inv, pm.rel, 1);
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ch = (tid >> 7) + 2 * h, idx = tid & 127;
        const int gi = (idx < 64) ? 64 * I + idx : 64 * J + idx - 64;
        int val = -1;
#pragma unroll
        for (int cc = 0; cc < MAXCH; ++cc)
          if (cc == ch && cc < px.n && gi < u && !ghost) val = px.inv[cc][w + gi];
        invs[tid + 256 * h] = val;
      }
      __syncthreads();
      if (!idle) schur_tile_gather(px, invs, i0, j0, li, lk, uv);
      nx = px.next;
    }
  }
  if (idle) return;
  // acc[x][y][q] = update of U(i = 64 I + i0 + 16 y + li, j = 64 J + j0 + 16 x + lk + 4 q)
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int i = 64 * I + i0 + 16 * y + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 64 * J + j0 + 16 * x + lk + 4 * q;
        if (i < u && j < u && i >= j) {
          c.Us[i + (long long)j * u] = uv[x][y][q] - acc[x][y][q];
        }
      }
    }
}

// ---- phase D on the narrow top levels of the single-launch factorisation: one 32 x 32 tile (I, J)
// per workgroup.  There a 64 x 64 tile is bound by the matrix pipes of ONE CU (336 MFMAs on four
// SIMDs) and by three staging round trips; with plenty of idle CUs the tile is cut in four, both
// operand strips (32 rows x w each) are requested in ONE batch right after the wait, and each of
// the waves 0..3 then owns one 16 x 16 block (w / 4 MFMAs).  Same products, same k order, same
// child order as dev_schur_tile: identical bits.  All threads of the workgroup stage.
// LDS (strips): 2 x wp x 32 doubles at c.A.
__device__ __forceinline__ void dev_schur_tile32(const FrontCtx& c, int I, int J, const PullCtx& pc,
                                                 int* __restrict__ wait_addr, int wait_target,
                                                 int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int w = c.w, r = c.r, u = c.u;
  const double* __restrict__ P21 = c.P + w;
  double* SI = c.A;             // [k][32 rows of I]
  double* SJ = c.A + 32 * c.wp;  // [k][32 rows of J] * d_k
  const int wi = wave & 1, wj = (wave >> 1) & 1;
  const bool mm = wave < 4 && !(I == J && wi < wj);  // this wave owns a block on or below the diagonal
  // children's entries of the block (they finished long ago): maps, then gathers, before the wait
  double uv[4] = {0.0, 0.0, 0.0, 0.0};
  const int gi = 32 * I + 16 * wi + li;
  if (mm) {
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      if (ch < pc.n) {
        const int ci = (gi < u) ? pc.inv[ch][w + gi] : -1;
        int cj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gj = 32 * J + 16 * wj + lk + 4 * q;
          cj[q] = (gj < u) ? pc.inv[ch][w + gj] : -1;
        }
        double g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = ci >= 0 && cj[q] >= 0 && ci >= cj[q];
          const double gv = pc.Uc[ch][ok ? ci + (long long)cj[q] * pc.uc[ch] : 0];
          g[q] = ok ? gv : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) uv[q] += g[q];
      }
  }
  TRW(5);
  flag_wait_ge(wait_addr, wait_target, info);
  TRW(1);
  // both strips in one batch: element e -> (k = e / 64, strip = (e / 32) & 1, row = e % 32)
  {
    double v[16], d[16];  // 64 wp <= 8192 elements
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = tid + 512 * t;
      const int k = e >> 6, row = e & 31;
      const bool js = (e >> 5) & 1;
      const int g = js ? 32 * J + row : 32 * I + row;
      v[t] = (k < w && g < u) ? P21[g + (long long)k * r] : 0.0;
      d[t] = (k < w && js) ? c.P[k + (long long)k * r] : 1.0;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = tid + 512 * t;
      const int k = e >> 6, row = e & 31;
      if (k < c.wp) {
        if ((e >> 5) & 1)
          SJ[k * 32 + row] = v[t] * d[t];
        else
          SI[k * 32 + row] = v[t];
      }
    }
  }
  __syncthreads();
  if (!mm) return;
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < w; k0 += 4) {
    const int kk = (k0 + lk) * 32;  // rows k >= w of the strips are zero (k < wp)
    acc = MFMA_F64(SJ[kk + 16 * wj + li], SI[kk + 16 * wi + li], acc);
  }
  if (gi < u) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gj = 32 * J + 16 * wj + lk + 4 * q;
      if (gj < u && gi >= gj) c.Us[gi + (long long)gj * u] = uv[q] - acc[q];
    }
  }
}

// fused: one workgroup per front, phases B, C, D (assembly has its own kernel)
__global__ __launch_bounds__(FB) void k_factor_level(const SnDesc* __restrict__ sn,
                                                     const int* __restrict__ level_sn, double* __restrict__ L,
                                                     double* __restrict__ U, const int* __restrict__ rel,
                                                     const int* __restrict__ child_idx, int* __restrict__ info,
                                                     int phases) {
  // phases: bit mask A(1) B(2) C(4) D(8); anything but 15 is a timing-only build
  // of the same kernel (results are then wrong by construction).
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const SnDesc S = sn[level_sn[blockIdx.x]];
  const FrontCtx c = make_ctx(S, L, U, lds);
  if (!(phases & 2)) return;
  PullCtx nopull;
  nopull.n = 0;
  dev_pivot_block<false>(c, info, phases, nopull, no_wait());
  dev_store_pivot_block(c);
  if (!(phases & 4)) return;
  {
    // scale by the reciprocal pivots like the split panel kernel (same bits): 1 / d_k into the free Y panel
    for (int k = threadIdx.x; k < c.wp; k += blockDim.x) c.Yp[k] = 1.0 / c.dd[k];
    __syncthreads();
    FrontCtx cr = c;
    cr.dd = c.Yp;
    dev_panel_solve<true, true>(cr, 0, 1, nopull, nullptr);
  }
  __syncthreads();
  if (c.u > 0 && (phases & 8)) {
    double* SI = c.A;
    double* SJ = c.A + 64 * KC;
    const int nt = (c.u + 63) >> 6;
    for (int I = 0; I < nt; ++I)
      for (int J = 0; J <= I; ++J) dev_schur_tile<true>(c, SI, SJ, I, J, S.child_begin == S.child_end, nopull, threadIdx.x);
  }
}

// phase A as its own kernel: items[2 * blockIdx.x] = supernode, items[2 * blockIdx.x + 1] = part
__global__ __launch_bounds__(1024) void k_front_assemble(const SnDesc* __restrict__ sn, const int* __restrict__ items,
                                                       int nparts, double* __restrict__ L, double* __restrict__ U,
                                                       const int* __restrict__ rel,
                                                       const int* __restrict__ child_idx) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const SnDesc S = sn[items[2 * blockIdx.x]];
  FrontCtx c = make_ctx(S, L, U, nullptr);
  dev_assemble(S, c, sn, U, rel, child_idx, items[2 * blockIdx.x + 1], nparts, reinterpret_cast<int*>(lds));
}

// split kernels B / C / D: one self-contained FrontItem per workgroup.  CHAIN: the level has fronts
// with more than MAXCH children (further descriptor blocks in `more`)
template <bool CHAIN>
__global__ __launch_bounds__(512) void k_front_pivot(const FrontItem* __restrict__ items, double* __restrict__ L,
                                                    double* __restrict__ U, int* __restrict__ info,
                                                    const int* __restrict__ inv, const int* __restrict__ rel,
                                                    const PullDesc* __restrict__ more, int pull) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const FrontItem& S = items[blockIdx.x];
  const FrontCtx c = make_ctx(S, L, U, lds);
  const PullCtx pc = make_pull(S.pd, U, inv, rel, pull);
#ifdef HIPFACT_PIVOT_V1
  dev_pivot_block<true, CHAIN>(c, info, 15, pc, no_wait(), PullMore{more, U, inv, rel});  // stores the finished tiles itself
#else
  dev_pivot_block<false, CHAIN, true>(c, info, 15, pc, no_wait(), PullMore{more, U, inv, rel});
  dev_store_pivot_block(c);
#endif
}

// LDS: dd | X | MAXCH x wp ints (the children's inverse maps of the pivot columns)
template <bool CHAIN>
__global__ __launch_bounds__(512) void k_front_panel(const FrontItem* __restrict__ items, double* __restrict__ L,
                                                    double* __restrict__ U, const int* __restrict__ inv,
                                                    const int* __restrict__ rel, const PullDesc* __restrict__ more,
                                                    int pull) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const FrontItem& S = items[blockIdx.x];
  const FrontCtx c = make_ctx(S, L, U, lds);
  const PullCtx pc = make_pull(S.pd, U, inv, rel, pull);
  int* invl = reinterpret_cast<int*>(c.A + c.wp * c.lda);
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (ch < pc.n)
      for (int k = threadIdx.x; k < c.wp; k += blockDim.x) invl[ch * c.wp + k] = (k < c.w) ? pc.inv[ch][k] : -1;
  // this wave's 16 panel rows leave for the registers before X is staged: one round trip for both
  const int R0 = c.w + 16 * (blockDim.x >> 6) * S.part + 16 * (threadIdx.x >> 6);
  const bool rok = (R0 + (int)(threadIdx.x & 15)) < c.r;
  double pv[8][4];
  int cis[MAXCH];
  dev_panel_rows_load(c, R0, pc, pv, cis);
  dev_load_pivot_block(c, true, true);
  dev_panel_rows_gather(c, pc, invl, pv, cis);
  // fronts with more than MAXCH children: further blocks of children, in child order
  for (int nx = CHAIN ? pc.next : -1; nx >= 0;) {
    const PullCtx px = make_pull(more[nx], U, inv, rel, 1);
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      for (int k = threadIdx.x; k < c.wp; k += blockDim.x) invl[ch * c.wp + k] = (ch < px.n && k < c.w) ? px.inv[ch][k] : -1;
      cis[ch] = (ch < px.n && rok) ? px.inv[ch][R0 + (threadIdx.x & 15)] : -1;
    }
    __syncthreads();
    dev_panel_rows_gather(c, px, invl, pv, cis);
    nx = px.next;
  }
  dev_panel_rows_product<true, true>(c, R0, pv);
}

// part = (I << 16) | J
template <bool CHAIN>
__global__ __launch_bounds__(FB, 3) void k_front_schur(const FrontItem* __restrict__ items, double* __restrict__ L,
                                                    double* __restrict__ U, const int* __restrict__ inv,
                                                    const int* __restrict__ rel, const PullDesc* __restrict__ more,
                                                    int pull) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const FrontItem& S = items[blockIdx.x];
  const FrontCtx c = make_ctx(S, L, U, lds);
  const PullCtx pc = make_pull(S.pd, U, inv, rel, pull);
  const int ij = S.part;
  dev_schur_tile<false, CHAIN>(c, c.A, c.A + 64 * KC, ij >> 16, ij & 0xffff, S.nchild == 0, pc, threadIdx.x, nullptr, 0,
                               nullptr, PullMore{more, U, inv, rel});
}

// ---------------------------------------------------------------------------
// Level-scheduled solves.  y holds the right-hand side in pivot order on entry
// and the solution on exit.  Forward: children -> parents, the coupling to the
// ancestors travels as update vectors (deterministic, no atomics).  Backward:
// parents -> children, gathers the ancestors' solution.
// ---------------------------------------------------------------------------
constexpr int SB = 1024;  // threads per block of the solve kernels (16 waves hide the panel-read latency)

// forward step of one front (all threads of the block; lds: r + 9 w + 1024 doubles)
__device__ __forceinline__ void dev_fwd_front(const SnDesc& S, const SnDesc* __restrict__ sn,
                                              const double* __restrict__ L, const int* __restrict__ rel,
                                              const int* __restrict__ child_idx, double* __restrict__ y,
                                              double* __restrict__ uvec, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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


__device__ __forceinline__ void top_wait(int* __restrict__ flags, int who, int* __restrict__ info, int target = 1);


// Forward step of one front inside the single-launch top-of-tree kernel.  Everything that does
// not depend on the children is requested BEFORE the wait for their flags: own right-hand side,
// the children's relative indices (LDS), this thread's fragments of inv(L11) (registers) and L21
// (LDS).  After the wait only the children's update vectors are one memory round trip away.
// Same arithmetic, in the same order, as dev_fwd_front.
__device__ __forceinline__ void dev_fwd_front_top(const TopItem& T, const double* __restrict__ L,
                                                  const int* __restrict__ inv, double* __restrict__ y,
                                                  double* __restrict__ uvec, double* lds, int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = T.w, r = T.r, u = r - w;
  const double* __restrict__ P = L + T.Loff;
  double* f = lds;            // r
  double* xs = f + r;         // w
  double* ps = xs + w;        // 8 x w partial sums of the triangular product
  double* part = ps + 8 * w;  // <= 1024 partial sums of the rectangular product
  double* Lb = part + 1024 + TOP_REL_CAP / 2;       // u x w, column-major
  // front row tid (r <= SB): own right-hand side and, per child, which of its update rows lands here
  double f0 = (tid < w) ? y[T.c0 + tid] : 0.0;
  int iv[MAXCH];
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) iv[ch] = (ch < T.nchild && tid < r) ? inv[T.c_invoff[ch] + tid] : -1;
  // row k of inv(L11), eighth p of the column range [0, k): at most 16 entries
  const int xk = tid & 127, xp = tid >> 7;
  const int xlo = (int)(((long long)xk * xp) >> 3), xhi = (int)(((long long)xk * (xp + 1)) >> 3);
  double xr[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) xr[j] = (xk < w && xlo + j < xhi) ? P[xk + (long long)(xlo + j) * r] : 0.0;
  for (int k0 = 4 * wave; k0 < w; k0 += 64)
    for (int a = lane; a < u; a += 64) {
      double v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (k0 + c < w) ? P[w + a + (long long)(k0 + c) * r] : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (k0 + c < w) Lb[a + (k0 + c) * u] = v[c];
    }
  // ---- the children's contributions, polled element by element, added in child order
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (iv[ch] >= 0) f0 += poll_f64(uvec + T.c_uoff[ch] + iv[ch], info);
  if (tid < r) f[tid] = f0;
  __syncthreads();
  if (xk < w) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int n = xhi - xlo, n4 = n & ~3;
#pragma unroll
    for (int j = 0; j < 16; j += 4)
      if (j < n4) {
        s0 += xr[j] * f[xlo + j];
        s1 += xr[j + 1] * f[xlo + j + 1];
        s2 += xr[j + 2] * f[xlo + j + 2];
        s3 += xr[j + 3] * f[xlo + j + 3];
      }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j >= n4 && j < n) s0 += xr[j] * f[xlo + j];
    ps[xp * w + xk] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  for (int k = tid; k < w; k += SB) {
    double s = f[k];
#pragma unroll
    for (int p = 0; p < 8; ++p) s += ps[p * w + k];
    xs[k] = s;
    y[T.c0 + k] = s;
  }
  __syncthreads();
  if (u > 0) {
    const int nchunk = (u + 63) >> 6;
    const int nslice = nchunk >= 16 ? 1 : 16 / nchunk;  // u * w <= TOP_L21_CAP: nchunk < 16 unless w < 17
    double* __restrict__ us = uvec + T.uoff;
    if (nslice == 1) {
      for (int ch = wave; ch < nchunk; ch += 16) {
        const int a = (ch << 6) + lane;
        if (a < u) {
          const double* Lr = Lb + a;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
          int k = 0;
          for (; k + 3 < w; k += 4) {
            s0 += Lr[k * u] * xs[k];
            s1 += Lr[(k + 1) * u] * xs[k + 1];
            s2 += Lr[(k + 2) * u] * xs[k + 2];
            s3 += Lr[(k + 3) * u] * xs[k + 3];
          }
          for (; k < w; ++k) s0 += Lr[k * u] * xs[k];
          post_f64(us + a, f[w + a] - ((s0 + s1) + (s2 + s3)));
        }
      }
    } else {
      const int ch = wave % nchunk, sl = wave / nchunk;
      if (sl < nslice) {
        const int a = (ch << 6) + lane;
        const int lo = (int)(((long long)w * sl) / nslice), hi = (int)(((long long)w * (sl + 1)) / nslice);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (a < u) {
          const double* Lr = Lb + a;
          int k = lo;
          for (; k + 3 < hi; k += 4) {
            s0 += Lr[k * u] * xs[k];
            s1 += Lr[(k + 1) * u] * xs[k + 1];
            s2 += Lr[(k + 2) * u] * xs[k + 2];
            s3 += Lr[(k + 3) * u] * xs[k + 3];
          }
          for (; k < hi; ++k) s0 += Lr[k * u] * xs[k];
        }
        part[sl * (nchunk << 6) + (ch << 6) + lane] = (s0 + s1) + (s2 + s3);
      }
      __syncthreads();
      for (int a = tid; a < u; a += SB) {
        double s = 0.0;
        for (int sl2 = 0; sl2 < nslice; ++sl2) s += part[sl2 * (nchunk << 6) + a];
        post_f64(us + a, f[w + a] - s);
      }
    }
  }
}

__global__ __launch_bounds__(SB) void k_fwd_level(const SnDesc* __restrict__ sn, const int* __restrict__ level_sn,
                                                  const double* __restrict__ L, const int* __restrict__ rel,
                                                  const int* __restrict__ child_idx, double* __restrict__ y,
                                                  double* __restrict__ uvec, const int* __restrict__ skip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (skip && *skip) return;
  const SnDesc S = sn[level_sn[blockIdx.x]];
  dev_fwd_front(S, sn, L, rel, child_idx, y, uvec, lds);
}


// backward step of one front (lds: u + w doubles).  In the single-launch top-of-tree
// kernel (flags != nullptr) the wait for the parent happens AFTER the panel
// fragments have been requested, so their latency overlaps the dependency wait.
__device__ __forceinline__ void dev_bwd_front(const SnDesc& S, const double* __restrict__ L,
                                              const int* __restrict__ rows, double* __restrict__ y, double* lds,
                                              int* __restrict__ flags = nullptr, int* __restrict__ info = nullptr,
                                              double* __restrict__ ysol = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = S.w, r = S.r, u = r - w;
  const double* __restrict__ P = L + S.Loff;
  const int* __restrict__ rw = rows + S.rowoff + w;
  double* g = lds;      // u
  double* v = lds + u;  // w
  // columns of this wave: k = 4 (wave + 16 p) + c, p = 0, 1 (w <= 128)
  const bool pre = (u <= 256);
  double lv[2][4][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = 4 * (wave + 16 * p) + c;
      const double* col = P + (long long)k * r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int a = lane + 64 * q;
        lv[p][c][q] = (pre && k < w && a < u) ? col[w + a] : 0.0;
      }
    }
  if (flags && S.parent >= 0) top_wait(flags, S.parent, info);  // the parent is done only after all its ancestors
  for (int a = tid; a < u; a += SB) g[a] = y[rw[a]];
  __syncthreads();
  // v_k = z_k / d_k - L21(:,k)^T g
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (4 * (wave + 16 * p) >= w) continue;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (pre) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = lane + 64 * q;
          if (a < u) s[c] += lv[p][c][q] * g[a];
        }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 4 * (wave + 16 * p) + c;
        if (k < w) {
          const double* col = P + w + (long long)k * r;
          for (int a = lane; a < u; a += 64) s[c] += col[a] * g[a];
        }
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
        const int k = 4 * (wave + 16 * p) + c;
        if (k < w) v[k] = y[S.c0 + k] / P[k + (long long)k * r] - s[c];
      }
    }
  }
  // fragments of inv(L11) for the second product (in flight across the barrier)
  double xv[2][4][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = 4 * (wave + 16 * p) + c;
      const double* col = P + (long long)k * r;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = k + 1 + lane + 64 * q;
        xv[p][c][q] = (k < w && t < w) ? col[t] : 0.0;
      }
    }
  __syncthreads();
  // x_k = v_k + inv(L11)(:,k)^T v below the diagonal
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (4 * (wave + 16 * p) >= w) continue;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = 4 * (wave + 16 * p) + c;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = k + 1 + lane + 64 * q;
        if (k < w && t < w) s[c] += xv[p][c][q] * v[t];
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
        const int k = 4 * (wave + 16 * p) + c;
        if (k < w) {
          y[S.c0 + k] = v[k] + s[c];
          if (ysol) post_f64(ysol + S.c0 + k, v[k] + s[c]);
        }
      }
    }
  }
}


// Backward step of a front with u <= 256 update rows, both products on the matrix cores: a GEMV
// as an MFMA whose B operand repeats the vector in all 16 columns wastes 15/16 of the flops,
// but it needs no cross-lane reduction at all (a wave-level shuffle tree costs ~500 VALU
// instructions per wave, and with 16 waves per CU that is microseconds on the critical path).
//   v = z / d - L21^T g      wave (kb, sp): 16 pivot columns x one slice of the update rows
//   x = v + strict_lower(inv(L11))^T v
// Slices are added in a fixed order through LDS.  TOP: part of the single-launch top-of-tree
// kernel; everything that does not depend on the ancestors (L21 fragments in registers,
// inv(L11), z / d and the row list in LDS) is requested BEFORE the wait for the parent.
// lds (doubles): 4 ceil(u/4) | wp + 4 | wp | 256 | ceil(u/2) | TOP: w w
template <bool TOP>
__device__ __forceinline__ void dev_bwd_small(long long Loff, long long rowoff, int c0, int w, int r, int parent,
                                              const double* __restrict__ L, const int* __restrict__ rows,
                                              double* __restrict__ y, double* lds, double* __restrict__ ysol,
                                              int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int u = r - w;
  const int nbk = (w + 15) >> 4, wp = nbk << 4;
  const int nsplit = 16 / nbk;  // >= 2
  const int kb = wave % nbk, sp = wave / nbk;
  const bool active = sp < nsplit;
  const int nac = (u + 3) >> 2;  // chunks of 4 update rows, <= 64
  const int c_lo = nac * sp / nsplit, c_hi = nac * (sp + 1) / nsplit;
  const int k = 16 * kb + li;
  const double* __restrict__ P = L + Loff;
  const int* __restrict__ rw = rows + rowoff + w;
  double* g = lds;                // 4 nac
  double* v = g + 4 * nac;        // wp + 4 (zero beyond w)
  double* ypre = v + wp + 4;      // z_k / d_k
  double* part = ypre + wp;       // nsplit x wp <= 256 partial sums
  int* rwb = reinterpret_cast<int*>(part + 256);
  double* Xb = part + 256 + ((u + 1) >> 1);  // TOP: w x w, zero on and above the diagonal
  int myrow = -1;  // TOP: update row tid (u <= 256 < SB) is polled by this thread
  for (int a = tid; a < 4 * nac; a += SB) {
    if (a < u) {
      if (TOP)
        myrow = rw[a];
      else
        rwb[a] = rw[a];
    } else {
      g[a] = 0.0;
    }
  }
  for (int t = tid; t < wp + 4; t += SB) {
    if (t < w)
      ypre[t] = y[c0 + t] / P[t + (long long)t * r];
    else
      v[t] = 0.0;
  }
  if (TOP) {
    for (int k0 = 4 * wave; k0 < w; k0 += 64)
      for (int t = lane; t < w; t += 64) {
        double x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = (k0 + c < w && t > k0 + c) ? P[t + (long long)(k0 + c) * r] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (k0 + c < w) Xb[t + (k0 + c) * w] = x[c];
      }
  }
  // (requested last: the staging loops above then run without these 64 registers live)
  double lv[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const int a = 4 * (c_lo + j) + lk;
    lv[j] = (active && c_lo + j < c_hi && k < w && a < u) ? P[w + a + (long long)k * r] : 0.0;
  }
  if (TOP) {
    // the ancestors' solution entries, polled one by one (no flag, no fence: see poll_f64)
    if (myrow >= 0) g[tid] = poll_f64(ysol + myrow, info);
  } else {
    __syncthreads();
    for (int a = tid; a < u; a += SB) g[a] = y[rwb[a]];
  }
  __syncthreads();
  if (active) {
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (c_lo + j < c_hi) acc = MFMA_F64(lv[j], g[4 * (c_lo + j) + lk], acc);
    if (li == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) part[sp * wp + 16 * kb + lk + 4 * q] = acc[q];
    }
  }
  // strict lower part of inv(L11)^T: chunks of 4 rows t below the first column of the block
  const int ntc = (w + 3) >> 2;
  const int cnt = ntc - 4 * kb;
  const int t_lo = 4 * kb + cnt * sp / nsplit, t_hi = 4 * kb + cnt * (sp + 1) / nsplit;
  double xf[16];
  if (!TOP) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int t = 4 * (t_lo + j) + lk;
      xf[j] = (active && t_lo + j < t_hi && k < w && t < w && t > k) ? P[t + (long long)k * r] : 0.0;
    }
  }
  __syncthreads();
  for (int t = tid; t < w; t += SB) {
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += part[q * wp + t];
    v[t] = ypre[t] - s;
  }
  __syncthreads();
  if (active) {
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (t_lo + j < t_hi) {
        const int t = 4 * (t_lo + j) + lk;
        const double xa = TOP ? ((k < w && t < w) ? Xb[t + k * w] : 0.0) : xf[j];
        acc = MFMA_F64(xa, v[t], acc);
      }
    if (li == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) part[sp * wp + 16 * kb + lk + 4 * q] = acc[q];
    }
  }
  __syncthreads();
  for (int t = tid; t < w; t += SB) {
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += part[q * wp + t];
    y[c0 + t] = v[t] + s;
    if (TOP) post_f64(ysol + c0 + t, v[t] + s);
  }
}

__global__ __launch_bounds__(SB) void k_bwd_level(const SnDesc* __restrict__ sn, const int* __restrict__ level_sn,
                                                  const double* __restrict__ L, const int* __restrict__ rows,
                                                  double* __restrict__ y, const int* __restrict__ skip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (skip && *skip) return;
  const SnDesc S = sn[level_sn[blockIdx.x]];
  if (S.r - S.w <= 256)
    dev_bwd_small<false>(S.Loff, S.rowoff, S.c0, S.w, S.r, S.parent, L, rows, y, lds, nullptr, nullptr);
  else
    dev_bwd_front(S, L, rows, y, lds);
}

// ---------------------------------------------------------------------------
// The elimination tree of the solve in ONE launch per direction: every front gets its own
// workgroup, indexed so that a front never waits for one dispatched after it (workgroups are
// dispatched in index order, so progress needs no co-residency).  Ordinary fronts exchange their
// vectors element by element (poll_f64 / post_f64: the data is its own flag).  Wide and generic
// fronts use one done-flag per front:
//   producer: all waves drain their stores, block barrier, one lane issues an
//             agent-scope release and then a relaxed agent-scope flag store;
//   consumer: one lane polls the flag (relaxed, agent scope, bounded spin), then
//             an agent-scope acquire, block barrier, plain loads.
// The flags of one sweep are cleared by the kernel of the other.  A spin that runs out
// sets INFO_TIMEOUT instead of hanging the GPU.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void top_wait(int* __restrict__ flags, int who, int* __restrict__ info, int target) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(&flags[who], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) {
        atomicAdd(&info[INFO_TIMEOUT], 1);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void top_publish_add(int* __restrict__ flags, int who) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&flags[who], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void top_publish(int* __restrict__ flags, int who) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&flags[who], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------
// Top of the elimination tree of the FACTORISATION in one launch.  The last levels hold a
// handful of fronts each; three launches per level are then pure launch + dependent latency.
// Here every pivot / panel / Schur work item of those levels is a workgroup of one grid, ordered
// level by level and pivot -> panel -> Schur inside a level.  Dependencies travel through
// counters with the agent-scope release / acquire protocol:
//   pivot(f)  waits for all Schur workgroups of each child of f   (ddone[child] == count)
//   panel(f)  waits for pivot(f)                                   (bdone[f] == 1)
//   Schur(f)  waits for all panel workgroups of f                  (cdone[f] == count)
// A workgroup only ever waits for lower-indexed ones, and workgroups are dispatched in index
// order, so progress needs no co-residency.  Same device code, same arithmetic, same bits as
// the per-level kernels.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void flag_wait_ge(int* __restrict__ addr, int target, int* __restrict__ info) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) {
        atomicAdd(&info[INFO_TIMEOUT], 1);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void flag_publish_add(int* __restrict__ addr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(addr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Role 3 (fused solve): the solve panels S = [X; -L21 X] of one front, ANY front of the tree.  The launch is bound
// by the critical path of the tree; from the level on where a level's workgroups no longer fill the chip, the
// host deals these items between the levels, as many as leave room for the level and its successor - they
// run on CUs that would otherwise idle (3.7 GFLOP of fp64 MFMA work for the whole tree, ~100 us as a launch of
// its own behind this one).  Fronts of this launch wait for their pivot and panel workgroups, all others were
// finished before the launch.
__device__ __forceinline__ void dev_build_solve_panel(const SolveItem& T, const double* __restrict__ L,
                                                      double* __restrict__ SPf, double* __restrict__ SPb, double* lds);

// LDS: the largest of the roles (pivot: dd | A | Y | scratch | maps; panel: dd | X | maps;
// Schur: dd | two teams of {SI, SJ, maps}; solve panels: X | 1 / d | tiles | offsets)
__global__ __launch_bounds__(512) void k_factor_top(const TopFItem* __restrict__ items, double* __restrict__ L,
                                                   double* __restrict__ U, int* __restrict__ info,
                                                   const int* __restrict__ inv, const int* __restrict__ rel,
                                                   int* __restrict__ bdone, int* __restrict__ cdone,
                                                   int* __restrict__ ddone, double* __restrict__ xarena,
                                                   const SolveItem* __restrict__ sitems, double* __restrict__ SPf,
                                                   double* __restrict__ SPb, int zero_behind) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const TopFItem& T = items[blockIdx.x];
  const FrontItem& S = T.it;
  TRW(0);
#ifdef HIPFACT_TRACE
  if (threadIdx.x == 0 && blockIdx.x < TRACE_WGS) g_trace[blockIdx.x * 8 + 7] = T.role * 100000 + T.front;
#endif
  if (T.role == 3) {
    if (T.nwait) {  // a front of this launch: pivot block and every panel workgroup
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(&bdone[T.front], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1 ||
               __hip_atomic_load(&cdone[T.front], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T.target) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 22)) {
            atomicAdd(&info[INFO_TIMEOUT], 1);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    TRW(1);
    dev_build_solve_panel(sitems[S.part], L, SPf, SPb, lds);
    // A front BELOW this launch: nobody reads its panel any more (its parent took the update matrix, the solves
    // run on the solve panels) - it goes back to zero here, in the shadow of the latency-bound top of the tree,
    // instead of in the zero fill in front of the next factorisation.
    if (zero_behind && !T.nwait) {
      __syncthreads();
      double2* __restrict__ pz = reinterpret_cast<double2*>(L + S.Loff);
      const long long n2 = ((long long)S.r * S.w + 1) >> 1;
      for (long long e = threadIdx.x; e < n2; e += blockDim.x) pz[e] = double2{0.0, 0.0};
    }
#ifdef HIPFACT_TRACE
    __syncthreads();
#endif
    TRW(2);
    return;
  }
  FrontCtx c = make_ctx(S, L, U, lds);
  // Levels that are bound by latency (fewer workgroups than CUs) post the pivot block tile by tile for
  // their panel workgroups; on wider levels hundreds of polling workgroups would flood the memory
  // system for no gain (those levels are bound by occupancy), so they keep the flag.
  c.Xa = T.post ? xarena + T.xoff : nullptr;
  const PullCtx pc = make_pull(S.pd, U, inv, rel, 1);
  ChildWait cw;
  cw.n = T.nwait;
  cw.info = info;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) {
    cw.addr[ch] = ddone + T.wait_id[ch];
    cw.target[ch] = T.wait_cnt[ch];
  }
  // In every role, whatever does not depend on the awaited workgroups is requested before the
  // wait, so that afterwards only the awaited data is one memory round trip away.
  if (T.role == 0) {
#ifdef HIPFACT_PIVOT_V1
    dev_pivot_block<true>(c, info, 15, pc, cw);  // waits for the children between its prefetch and its gathers
#else
    dev_pivot_block<false, false, true>(c, info, 15, pc, cw);  // waits for the children between its prefetch and its gathers
    dev_store_pivot_block(c);                                  // inv(L11) and the pivots: the factor's final form
#endif
    TRW(2);
    flag_publish_add(&bdone[T.front]);
    TRW(3);
    // A front without update rows (the root): its solve panel is X and the pivots, both still in LDS - written here
    // instead of by an item of its own that could only start now (same expressions, same bits)
    if (T.sidx > 0) {
      const SolveItem& Q = sitems[T.sidx - 1];
      const int w = c.w, r = c.r, lda = c.lda;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const long long TSf = (long long)r * Q.Qf, TSb = (long long)w * Q.Pb;
      double* __restrict__ sf = SPf + Q.spf;
      double* __restrict__ sb = SPb + Q.spb;
      for (int k = wave; k < w; k += 8) {
        const long long of = (long long)(k / Q.Qf) * TSf + (long long)(k % Q.Qf) * r;
        for (int i = k + lane; i < w; i += 64) sf[of + i] = (i == k) ? 1.0 : c.A[i + k * lda];
      }
      for (int i = wave; i < w; i += 8) {
        const double di = 1.0 / c.dd[i];
        const long long ob = (long long)(i / Q.Pb) * TSb + (long long)(i % Q.Pb) * w;
        for (int k = lane; k <= i; k += 64) sb[ob + k] = ((i == k) ? 1.0 : c.A[i + k * lda]) * di;
      }
    }
  } else if (T.role == 1) {
    // panel rows incl. the children's contributions first (the children finished long ago),
    // then the pivot workgroup of the own front is awaited and only inv(L11) remains to be read
    cw.wait();
    int* invl = reinterpret_cast<int*>(c.A + c.wp * c.lda);
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      if (ch < pc.n)
        for (int k = threadIdx.x; k < c.wp; k += blockDim.x) invl[ch * c.wp + k] = (k < c.w) ? pc.inv[ch][k] : -1;
    // crows = 128: one 16-row strip per wave; 64: two waves per strip, alternating output blocks
    const int wv = threadIdx.x >> 6;
    const int cstep = T.prows == 64 ? 2 : 1;
    const int R0 = c.w + T.prows * S.part + 16 * (cstep == 2 ? (wv >> 1) : wv);
    double pv[8][4];
    int cis[MAXCH];
#ifndef HIPFACT_PIVOT_V1
    if (c.Xa) {  // the pivot workgroup posts L11: block substitution, one wave per strip of 16 rows
      const bool mine = 16 * wv < T.prows;  // (prows = 64: the waves 4 .. 7 only poll and stage)
      const int R1 = c.w + T.prows * S.part + 16 * wv;
      const int Rs = mine ? R1 : c.r;       // (an idle wave loads nothing)
      dev_panel_rows_load<true>(c, Rs, pc, pv, cis);
      __syncthreads();
      dev_panel_rows_gather<true>(c, pc, invl, pv, cis);
      TRW(1);
      dev_panel_rows_subst_posted(c, Rs, pv, mine && R1 < c.r, S.part == 0, info);
      TRW(2);
      flag_publish_add(&cdone[T.front]);
      TRW(3);
      return;
    }
#endif
    dev_panel_rows_load(c, R0, pc, pv, cis);
    __syncthreads();
    dev_panel_rows_gather(c, pc, invl, pv, cis);
    if (c.Xa) {
      dev_panel_rows_product_posted(c, R0, pv, cstep, cstep == 2 ? (wv & 1) : 0, R0 < c.r, info);
    } else {
      flag_wait_ge(&bdone[T.front], 1, info);
      dev_load_pivot_block(c, true, true);
      if (R0 < c.r) dev_panel_rows_product<true, true>(c, R0, pv, cstep, cstep == 2 ? (wv & 1) : 0);
    }
    flag_publish_add(&cdone[T.front]);
  } else {
    // two 256-thread teams, one tile each (the same tile twice when the front has an odd number):
    // children's entries first, then the panel workgroups of the own front are awaited
    cw.wait();
    if (T.crows == 64) {
      dev_schur_tile32(c, S.part >> 16, S.part & 0xffff, pc, &cdone[T.front], T.target, info);
    } else {
      const int team = threadIdx.x >> 8;
      const int ij = team ? T.part2 : S.part;  // part2 < 0: the second team has no tile
      double* SI = c.A + (size_t)team * (2 * 64 * KC + 64 * MAXCH);
      dev_schur_tile<false>(c, SI, SI + 64 * KC, ij < 0 ? -1 : (ij >> 16), ij & 0xffff, S.nchild == 0, pc,
                            threadIdx.x & 255, &cdone[T.front], T.target, info);
    }
    // every panel workgroup of the front has finished polling: its slot of posted tiles goes back
    // to the sentinel for the next factorisation (a share per Schur workgroup)
    if (c.Xa)
      for (int e = T.sidx * 512 + threadIdx.x; e < c.wp * c.wp; e += T.scount * 512) sent_f64(c.Xa + e);
    TRW(2);
    flag_publish_add(&ddone[T.front]);
    TRW(3);
  }
}

// ---- wide fronts in the single-launch solves.  A front with thousands of update rows streams
// megabytes of L21 per solve; one workgroup moves that at the bandwidth of one CU.  Such a front
// is split into a head (pivot block: the two small triangular products) and slices of
// WIDE_SLICE_ROWS update rows (the rectangular products), one workgroup each:
//   forward   head: f_top += children, x = inv(L11) f_top            -> hflags[F] = 1
//             slice: waits for the head; u[a] = f_below[a] - L21[a, :] x   -> flags[F] += 1   (F done at nsl)
//   backward  slice: waits for the parent; partial_s = L21[slice, :]^T g   -> hflags[F] += 1
//             head: waits for its slices; v = z / d - sum_s partial_s (fixed order), x = v + lower(inv(L11))^T v
//                                                                     -> flags[F] = 1
// Children's contributions are gathered through the inverse relative indices (no scatter, child order).
// The head's share of inv(L11): thread (row k = tid & 127, segment p = tid >> 7 of eight) owns the entries
// t in [k p / 8, k (p + 1) / 8) of row k - at most 16 for w <= 128.  They do not depend on the children: requested
// BEFORE the children are awaited (sixteen strided global loads per thread behind the wait were 4 of the 5 us a
// head spent on the critical path of every level of a dense chain, scripts/timeline_solve.py).
__device__ __forceinline__ void dev_fwd_wide_head_prefetch(const TopItem& T, const double* __restrict__ L,
                                                           double (&xr)[16]) {
  const int tid = threadIdx.x;
  const int w = T.w, r = T.r;
  const double* __restrict__ P = L + T.Loff;
  const int k = tid & 127, p = tid >> 7;
  const int lo = (int)(((long long)k * p) >> 3), hi = (int)(((long long)k * (p + 1)) >> 3);
#pragma unroll
  for (int j = 0; j < 16; ++j) xr[j] = (k < w && lo + j < hi) ? P[k + (long long)(lo + j) * r] : 0.0;
}
__device__ __forceinline__ void dev_fwd_wide_head(const TopItem& T, const double* __restrict__ L,
                                                  const int* __restrict__ rel, double* __restrict__ y,
                                                  const double* __restrict__ uvec, double* lds,
                                                  const double (&xr)[16]) {
  const int tid = threadIdx.x;
  const int w = T.w, r = T.r;
  const double* __restrict__ P = L + T.Loff;
  double* f = lds;       // w
  double* ps = f + w;    // 8 x w partial sums
  for (int t = tid; t < w; t += SB) f[t] = y[T.c0 + t];
  __syncthreads();
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch)
    if (ch < T.nchild) {
      // the child's update rows that land in the pivot rows are its leading ones (rel is monotone)
      const int* __restrict__ rc = rel + T.c_reloff[ch];
      const double* __restrict__ uv = uvec + T.c_uoff[ch];
      for (int a = tid; a < min(T.c_uc[ch], w); a += SB) {
        const int q = rc[a];
        if (q < w) f[q] += uv[a];
      }
      __syncthreads();
    }
  {
    const int k = tid & 127, p = tid >> 7;
    if (k < w) {
      const int lo = (int)(((long long)k * p) >> 3), hi = (int)(((long long)k * (p + 1)) >> 3);
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      if (w <= 128) {
        // same association as the loop below: groups of four from lo, the remainder into s0
        const int n4 = (hi - lo) & ~3;
#pragma unroll
        for (int j = 0; j < 16; j += 4)
          if (j < n4) {
            s0 += xr[j] * f[lo + j];
            s1 += xr[j + 1] * f[lo + j + 1];
            s2 += xr[j + 2] * f[lo + j + 2];
            s3 += xr[j + 3] * f[lo + j + 3];
          }
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j >= n4 && lo + j < hi) s0 += xr[j] * f[lo + j];
      } else {
        const double* Xk = P + k;
        int t = lo;
        for (; t + 3 < hi; t += 4) {
          s0 += Xk[(long long)t * r] * f[t];
          s1 += Xk[(long long)(t + 1) * r] * f[t + 1];
          s2 += Xk[(long long)(t + 2) * r] * f[t + 2];
          s3 += Xk[(long long)(t + 3) * r] * f[t + 3];
        }
        for (; t < hi; ++t) s0 += Xk[(long long)t * r] * f[t];
      }
      ps[p * w + k] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
  for (int k = tid; k < w; k += SB) {
    double s = f[k];
#pragma unroll
    for (int p = 0; p < 8; ++p) s += ps[p * w + k];
    y[T.c0 + k] = s;
  }
}

// lds: w | WIDE_SLICE_ROWS | 1024
__device__ __forceinline__ void dev_fwd_wide_slice(const TopItem& T, const double* __restrict__ L,
                                                   const int* __restrict__ inv, const double* __restrict__ y,
                                                   double* __restrict__ uvec, double* lds, int* __restrict__ hflags,
                                                   int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = T.w, r = T.r;
  const int us = T.a1 - T.a0;  // <= WIDE_SLICE_ROWS
  const double* __restrict__ P = L + T.Loff + w + T.a0;
  double* xs = lds;
  double* fb = xs + w;
  double* part = fb + WIDE_SLICE_ROWS;
  // 64-row chunks x column slices; this thread's <= 32 entries of L21 and the children's
  // entries are requested before the head of the front is awaited
  const int nchunk = (us + 63) >> 6;  // <= 4
  const int nslice = 16 / nchunk;
  const int ch = wave % nchunk, sl = wave / nchunk;
  const int pa = (ch << 6) + lane;
  const int lo = (int)(((long long)w * sl) / nslice), hi = (int)(((long long)w * (sl + 1)) / nslice);
  double lv[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) lv[j] = (sl < nslice && pa < us && lo + j < hi) ? P[pa + (long long)(lo + j) * r] : 0.0;
  for (int a = tid; a < us; a += SB) {
    double s = 0.0;
#pragma unroll
    for (int c2 = 0; c2 < MAXCH; ++c2)
      if (c2 < T.nchild) {
        const int ia = inv[T.c_invoff[c2] + w + T.a0 + a];
        if (ia >= 0) s += uvec[T.c_uoff[c2] + ia];
      }
    fb[a] = s;
  }
  top_wait(hflags, T.s, info, 1);
  for (int k = tid; k < w; k += SB) xs[k] = y[T.c0 + k];
  __syncthreads();
  if (sl < nslice) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
      // (entries beyond the slice are zero)
      const int k = lo + j;
      s0 += lv[j] * xs[min(k, w - 1)];
      s1 += lv[j + 1] * xs[min(k + 1, w - 1)];
      s2 += lv[j + 2] * xs[min(k + 2, w - 1)];
      s3 += lv[j + 3] * xs[min(k + 3, w - 1)];
    }
    part[sl * (nchunk << 6) + (ch << 6) + lane] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  double* __restrict__ uo = uvec + T.uoff + T.a0;
  for (int a = tid; a < us; a += SB) {
    double s = 0.0;
    for (int sl2 = 0; sl2 < nslice; ++sl2) s += part[sl2 * (nchunk << 6) + a];
    uo[a] = fb[a] - s;
  }
}

// lds: WIDE_SLICE_ROWS
__device__ __forceinline__ void dev_bwd_wide_slice(const TopItem& T, const double* __restrict__ L,
                                                   const int* __restrict__ rows, const double* __restrict__ ysol,
                                                   double* __restrict__ wpart, double* lds, int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = T.w, r = T.r;
  const int us = T.a1 - T.a0;
  const double* __restrict__ P = L + T.Loff + w + T.a0;
  const int* __restrict__ rw = rows + T.rowoff + w + T.a0;
  double* g = lds;
  int* rwb = reinterpret_cast<int*>(g + WIDE_SLICE_ROWS);
  // wave: 8 columns, lanes: rows lane + 64 q.  The slice of L21 (32 entries per thread) and the row
  // list are requested before the parent is awaited.
  double lv[8][WIDE_SLICE_ROWS / 64];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int k = 8 * wave + cc;
    const double* col = P + (long long)k * r;
#pragma unroll
    for (int q = 0; q < WIDE_SLICE_ROWS / 64; ++q) {
      const int a = lane + 64 * q;
      lv[cc][q] = (k < w && a < us) ? col[a] : 0.0;
    }
  }
  // the ancestors' solution entries of the slice's rows are polled in the posted copy of the solution (every
  // backward item posts its entries there): no flag of the parent, no fence - the slice goes ahead as soon as
  // ITS rows are there, which for all but the first slice of a chain front is long before the parent is done
  for (int a = tid; a < WIDE_SLICE_ROWS; a += SB) {
    const int row = (a < us) ? rw[a] : -1;
    g[a] = (row >= 0) ? poll_f64(ysol + row, info) : 0.0;
  }
  (void)rwb;
  __syncthreads();
  // shuffle tree per column (fixed order)
  double s[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    s[cc] = 0.0;
#pragma unroll
    for (int q = 0; q < WIDE_SLICE_ROWS / 64; ++q) s[cc] += lv[cc][q] * g[lane + 64 * q];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) s[cc] += __shfl_down(s[cc], o, 64);
  }
  if (lane == 0) {
    const int sidx = T.a0 / WIDE_SLICE_ROWS;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int k = 8 * wave + cc;
      if (k < w) post_f64(wpart + T.poff + (long long)sidx * w + k, s[cc]);  // polled by the head
    }
  }
}

// lds: 2 w + nsl w
__device__ __forceinline__ void dev_bwd_wide_head(const TopItem& T, const double* __restrict__ L,
                                                  double* __restrict__ y, double* __restrict__ wpart,
                                                  double* lds, int* __restrict__ flags, int* __restrict__ hflags,
                                                  int* __restrict__ info, double* __restrict__ ysol) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = T.w, r = T.r;
  const double* __restrict__ P = L + T.Loff;
  double* v = lds;
  double* zd = v + w;  // z_k / d_k
  // requested before the slices (or the parent) are awaited: inv(L11) fragments, z / d
  double xv[8][2];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int k = 8 * wave + cc;
    const double* col = P + (long long)k * r;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = k + 1 + lane + 64 * q;
      xv[cc][q] = (k < w && t < w) ? col[t] : 0.0;
    }
  }
  for (int k = tid; k < w; k += SB) zd[k] = y[T.c0 + k] / P[k + (long long)k * r];
  // the slices' partial sums are their own flags: posted element by element, polled here (all of them at once,
  // one or a few per thread), staged in LDS and added in slice order there; the slots go back to the sentinel
  // for the next solve (single consumer).  A flag hop instead costs the slices a release (3 us: it writes the
  // L2 back) and the head a second round trip.
  double* pst = zd + w;  // nsl x w
  for (int e = tid; e < T.nsl * w; e += SB) {
    pst[e] = poll_f64(wpart + T.poff + e, info);
    sent_f64_agent(wpart + T.poff + e);
  }
  if (T.nsl == 0 && T.parent >= 0)
    top_wait(flags, T.parent, info, 1);
  else
    __syncthreads();
  for (int k = tid; k < w; k += SB) {
    double s = 0.0;
    for (int q = 0; q < T.nsl; ++q) s += pst[q * w + k];
    v[k] = zd[k] - s;
  }
  __syncthreads();
  // x_k = v_k + inv(L11)(:,k)^T v below the diagonal: wave: 8 columns, lanes: rows k + 1 + lane + 64 q
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int k = 8 * wave + cc;
    double s = 0.0;
    if (k < w) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = k + 1 + lane + 64 * q;
        if (t < w) s += xv[cc][q] * v[t];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0 && k < w) {
      y[T.c0 + k] = v[k] + s;
      post_f64(ysol + T.c0 + k, v[k] + s);
    }
  }
}

// titems: fronts of the levels >= top_level, children before parents
__global__ __launch_bounds__(SB) void k_fwd_top(const SnDesc* __restrict__ sn, const TopItem* __restrict__ titems,
                                                int top_level, const double* __restrict__ L,
                                                const int* __restrict__ rel, const int* __restrict__ child_idx,
                                                const int* __restrict__ inv, const int* __restrict__ ftarget,
                                                double* __restrict__ y, double* __restrict__ uvec,
                                                int* __restrict__ flags, int* __restrict__ hflags,
                                                int* __restrict__ info, int* __restrict__ stale, int nstale,
                                                double* __restrict__ ysol, const int* __restrict__ skip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (skip && *skip) return;  // whole launch (and the backward one with it): the slot / flag state stays consistent
  // the flags of the opposite sweep are idle while this kernel runs: clear them for its next launch
  for (int i = blockIdx.x * SB + threadIdx.x; i < nstale; i += gridDim.x * SB) stale[i] = 0;
  const TopItem& T = titems[blockIdx.x];
  // ... and so is the polled copy of the solution: the front's pivot entries go back to the sentinel
  if (T.kind != 2)
    for (int k = threadIdx.x; k < T.w; k += SB) sent_f64(ysol + T.c0 + k);
  if (T.kind != 0) {
    double xr[16];
    if (T.kind == 1) dev_fwd_wide_head_prefetch(T, L, xr);
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      if (ch < T.nchild && T.c_wait[ch]) top_wait(flags, T.c_id[ch], info, T.c_wait[ch]);
    if (T.kind == 1) {
      dev_fwd_wide_head(T, L, rel, y, uvec, lds, xr);
      top_publish(hflags, T.s);
    } else {
      dev_fwd_wide_slice(T, L, inv, y, uvec, lds, hflags, info);  // awaits the head after its prefetch
      top_publish_add(flags, T.s);
    }
    return;
  }
  if (T.prefetch & 1) {
    dev_fwd_front_top(T, L, inv, y, uvec, lds, info);
  } else {
    const SnDesc S = sn[T.s];
    for (int ci = S.child_begin; ci < S.child_end; ++ci) {
      const int c = child_idx[ci];
      if (sn[c].pad0 >= top_level) top_wait(flags, c, info, ftarget[c]);
    }
    dev_fwd_front(S, sn, L, rel, child_idx, y, uvec, lds);
  }
  top_publish(flags, T.s);
}

__global__ __launch_bounds__(SB) void k_bwd_top(const SnDesc* __restrict__ sn, const TopItem* __restrict__ titems,
                                                const double* __restrict__ L, const int* __restrict__ rows,
                                                double* __restrict__ y, double* __restrict__ wpart,
                                                int* __restrict__ flags, int* __restrict__ hflags,
                                                int* __restrict__ info, int* __restrict__ stale, int nstale,
                                                double* __restrict__ ysol, double* __restrict__ uvec,
                                                const int* __restrict__ skip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (skip && *skip) return;
  for (int i = blockIdx.x * SB + threadIdx.x; i < nstale; i += gridDim.x * SB) stale[i] = 0;
  // parents before children: workgroups are dispatched in index order, so a front never waits
  // for one that has not been dispatched yet (no co-residency assumption for correctness)
  const TopItem& T = titems[gridDim.x - 1 - blockIdx.x];
  // the update vector of the front was consumed by its parent in the forward launch: back to the sentinel
  if (T.kind == 0)
    for (int a = threadIdx.x; a < T.r - T.w; a += SB) sent_f64(uvec + T.uoff + a);
  else if (T.kind == 2)
    for (int a = T.a0 + threadIdx.x; a < T.a1; a += SB) sent_f64(uvec + T.uoff + a);
  if (T.kind == 2) {
    dev_bwd_wide_slice(T, L, rows, ysol, wpart, lds, info);  // polls its ancestors' entries after its prefetch
    return;
  }
  if (T.kind == 1) {
    dev_bwd_wide_head(T, L, y, wpart, lds, flags, hflags, info, ysol);  // awaits its slices after its prefetch
    top_publish(flags, T.s);
    return;
  }
  if (T.prefetch & 2) {
    dev_bwd_small<true>(T.Loff, T.rowoff, T.c0, T.w, T.r, T.parent, L, rows, y, lds, ysol, info);
  } else {
    const SnDesc S = sn[T.s];
    dev_bwd_front(S, L, rows, y, lds, flags, info, ysol);
  }
  top_publish(flags, T.s);
}

// ---------------------------------------------------------------------------
// The whole elimination tree of a solve in ONE launch (fused forward + backward sweep) on the
// "solve panels" S = [X; -W], X = inv(L11), W = L21 X.
//
// Why another form of the factor.  With L21 itself a front's forward step is two dependent products
// (x^ = X f_top, then u = f_below - L21 x^) and so is its backward step; every product is a
// barrier-separated phase of a 1024-thread workgroup, and the tree's critical path pays them level
// after level.  With W = L21 X both sweeps are ONE product per front:
//     forward   [x^; u] = [X; -W] f_top + [0; f_below]
//     backward  x = [X; -W]^T [D^-1 x^; g]          (since X^T L21^T = W^T)
// The panels are stored twice, thread-major (SolveItem), so that a thread's share of the front sits
// in registers BEFORE its dependency wait and the product afterwards is a run of register FMAs
// against a vector in LDS; the only cross-thread step is the sum of Qf (Pb) partials through LDS.
// 2 x nfronts workgroups: blocks [0, nf) forward, children before parents; blocks [nf, 2 nf)
// backward, parents before children.  Workgroups are dispatched in index order and only ever wait
// for lower-indexed ones, so progress needs no co-residency.  Data is its own flag (poll_f64 /
// post_f64): update vectors (forward), x^ (forward item -> backward item of the same front: this is
// how the root turns around without a launch boundary), solution copy ysol (backward).  Every
// polled slot is put back to the sentinel by its consumer's side: uvec and x^ by the backward item
// of the front (agent-scope stores: other slots of the same cache lines are live in this launch).
// ysol exists twice and the launches alternate (`epoch`, advanced by the kernel that follows the
// tree): a backward item may start polling long before the forward item of an ancestor has run,
// so it must never find the previous solve's value - the copy of the previous launch is put back to
// the sentinel by the forward items while this launch exchanges through the other one.
// ---------------------------------------------------------------------------
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
};
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }
// Reduces the partial maxima a residual kernel has left and updates the control block: one workgroup (any size
// that is a multiple of 64, at most 1024 threads).  (An election of the last block inside the residual kernel
// costs more: thousands of blocks each end with a dependent store -> ticket round trip, and increments of one
// word serialise.)
struct DecideIn {
  RefineCtl* ctl;
  RefineCtl* hctl;  // pinned copy for the host
  const double* partials;
  int nblk;
  double target;
  const unsigned long long* minmax;
};
__device__ __forceinline__ void dev_refine_decide(const DecideIn& D, int first) {
  RefineCtl* __restrict__ ctl = D.ctl;
  RefineCtl* __restrict__ hctl = D.hctl;
  __shared__ double sh[3][16];
  const int tid = threadIdx.x, nthr = blockDim.x;
  double r = 0.0, bb = 0.0, zz = 0.0;
  for (int q = tid; q < D.nblk; q += nthr) {
    r = nanmax(r, D.partials[3 * q]);
    bb = nanmax(bb, D.partials[3 * q + 1]);
    zz = nanmax(zz, D.partials[3 * q + 2]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    r = nanmax(r, __shfl_down(r, o, 64));
    bb = nanmax(bb, __shfl_down(bb, o, 64));
    zz = nanmax(zz, __shfl_down(zz, o, 64));
  }
  if ((tid & 63) == 0) {
    sh[0][tid >> 6] = r;
    sh[1][tid >> 6] = bb;
    sh[2][tid >> 6] = zz;
  }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < nthr / 64; ++q) {
      r = nanmax(r, sh[0][q]);
      bb = nanmax(bb, sh[1][q]);
      zz = nanmax(zz, sh[2][q]);
    }
    const unsigned long long* minmax = D.minmax;
    const double target = D.target;
    const double lo = __longlong_as_double((long long)~minmax[0]), hi = __longlong_as_double((long long)minmax[1]);
    const double kappa = (minmax[1] == 0ull) ? 10.0 : ((lo > 0.0 && hi >= lo) ? 10.0 * hi / lo : 1e300);
    const double tol = fmin(1e-12, fmax(4.5e-16, target / kappa));
    const double den = zz + bb;
    const double omega = (r == 0.0) ? 0.0 : r / den;  // den == 0 with r != 0 cannot happen; NaN stays NaN
    const int iters = first ? 0 : ctl->iters + 1;
    const int seq = ctl->seq + (first ? 1 : 0);
    const double prev = first ? 1.7e308 : ctl->omega;
    int done = 0, status = 3;
    if (!(omega == omega) || !(den < 1.7e308)) {
      done = 1;
      status = 2;
    } else if (target < 0.0) {
      // non-adaptive mode: the passes of the graph run unconditionally
    } else if (omega <= tol) {
      done = 1;
      status = 0;
    } else if (!first && omega > 0.5 * prev) {
      done = 1;
      status = 1;
    }
    ctl->iters = iters;
    ctl->omega_prev = prev;
    ctl->omega = omega;
    ctl->tol = tol;
    ctl->kappa = kappa;
    ctl->status = status;
    ctl->seq = seq;
    ctl->pending = 0;
    ctl->done = done;  // read by the kernels of the next pass: a kernel boundary away
    // copy for the host in pinned memory (visible after the stream has been synchronised): no copy node
    hctl->iters = iters;
    hctl->omega_prev = prev;
    hctl->omega = omega;
    hctl->tol = tol;
    hctl->kappa = kappa;
    hctl->status = status;
    hctl->done = done;
    // last, relaxed: the stores of one thread to host memory arrive in order (the host may peek without a sync)
    __hip_atomic_store(&hctl->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// The verdict as a launch of its own.  pending_only: the deferred verdict of a solve whose graph carries none
// (the first pass of a solve without correction passes; normally picked up by the next solve's tree launch) -
// nothing happens if it has been delivered already.
__global__ __launch_bounds__(FB) void k_refine_decide(DecideIn D, int first, int pending_only) {
  if (pending_only) {
    if (!D.ctl->pending) return;
  } else if (!first && D.ctl->done) {
    return;
  }
  dev_refine_decide(D, first);
}

// (working-set maps and equilibration of the saddle-point front end: described with the saddle kernels below)
struct SaddleMaps {
  const int* __restrict__ vmap;
  const int* __restrict__ cmap;
  const double* __restrict__ dscale;  // per pivot position
  int n;
};
__device__ __forceinline__ int ext_row(const SaddleMaps& M, int s) { return M.cmap ? M.cmap[s] : M.n + s; }
constexpr int RL = 16;  // lanes per row of A^_p in the right-hand side product
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
// one row of t, RL lanes per row (k_rhs_saddle and the forward items: same partial sums, same shuffle tree, same bits)
__device__ __forceinline__ double rhs_row(const RhsIn& R, int k, int sub) {
  double s = 0.0;
  const int p1 = R.Ar_ptr[k + 1];
  if (R.M.vmap) {
    for (int p = R.Ar_ptr[k] + sub; p < p1; p += RL) {
      const int j = R.Ar_col[p], v = R.M.vmap[j];
      s += R.Ar_val[p] * R.b[v >= 0 ? v : j];
    }
  } else {
    for (int p = R.Ar_ptr[k] + sub; p < p1; p += RL) s += R.Ar_val[p] * R.b[R.Ar_col[p]];
  }
#pragma unroll
  for (int o = RL / 2; o > 0; o >>= 1) s += __shfl_down(s, o, RL);
  if (sub == 0) {
    const int i = ext_row(R.M, R.perm[k]);
    s = i >= 0 ? s - R.b[i] * R.M.dscale[k] : 0.0;
  }
  return s;
}
constexpr int ST = 1024;  // threads per workgroup

__device__ __forceinline__ void dev_solve_fwd(const SolveItem& T, const double* __restrict__ SPf,
                                              const long long* __restrict__ xuoff, const int* __restrict__ xinvoff,
                                              const int* __restrict__ inv, const double* __restrict__ y,
                                              double* __restrict__ xhat, double* __restrict__ uvec,
                                              double* __restrict__ ysol, double* lds, int* __restrict__ info,
                                              const RhsIn& R) {
  const int tid = threadIdx.x;
  const int w = T.w, Q = T.Qf, E = T.Ef;
  // rows of this item: the front's pivot rows (always staged: the product's input) and its update rows
  // [a0, a1); the item's panel copy holds the rows it puts out: the pivot rows only in slice 0
  const int a0 = T.a0, nu = T.a1 - T.a0;
  const int top = (T.sl == 0) ? w : 0;
  const int ro = top + nu;  // output rows
  const int rl = w + nu;    // staged rows
  const int TS = ro * Q;
  const bool active = tid < TS;
  const int q = active ? tid / ro : 0;
  double* f = lds;          // rl
  double* part = lds + rl;  // Q * ro <= 1024
  // the copy of the solution the PREVIOUS launch exchanged through: back to the sentinel
  if (tid < w) sent_f64(ysol + T.c0 + tid);
  const double* __restrict__ sp = SPf + T.spf + tid;
  double pv[SOLVE_PREFETCH];
#pragma unroll
  for (int e = 0; e < SOLVE_PREFETCH; ++e) pv[e] = (active && e < E) ? sp[(long long)e * TS] : 0.0;
  // staged row tid (front row fr): own right-hand side and, per child, which of its update rows lands here
  const int fr = tid + (tid >= w ? a0 : 0);
  double f0 = 0.0;
  if (R.Ar_ptr) {
    // the front's own rows of t, formed here instead of by a launch in front of this one (every item does this
    // at once when the launch starts; staged through the buffer of the partial sums, free until the product)
    for (int row = tid / RL; row < w; row += ST / RL) {
      const double s = rhs_row(R, T.c0 + row, tid % RL);
      if (tid % RL == 0) part[row] = s;
    }
    __syncthreads();
    if (tid < w) f0 = part[tid];
  } else if (tid < w) {
    f0 = y[T.c0 + tid];
  }
  int iv[MAXCH];
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) iv[ch] = (ch < T.nchild && tid < rl) ? inv[T.c_invoff[ch] + fr] : -1;
  {
    // all children's entries of this row requested at once (several children usually reach the same
    // separator rows: one round trip instead of one per child), re-polled only where still pending,
    // added in child order (deterministic)
    unsigned long long bits[MAXCH];
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      bits[ch] = iv[ch] >= 0 ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(uvec + T.c_uoff[ch] + iv[ch]),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : 0ull;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch)
      if (iv[ch] >= 0) {
        if (bits[ch] == SOLVE_SENT) bits[ch] = (unsigned long long)__double_as_longlong(poll_f64(uvec + T.c_uoff[ch] + iv[ch], info));
        f0 += __longlong_as_double((long long)bits[ch]);
      }
  }
  for (int x = T.xbegin; x < T.xend; ++x) {  // fronts with more than MAXCH children (rare)
    const int ia = (tid < rl) ? inv[xinvoff[x] + fr] : -1;
    if (ia >= 0) f0 += poll_f64(uvec + xuoff[x] + ia, info);
  }
  if (tid < rl) f[tid] = f0;
  __syncthreads();
  double acc = 0.0;
#pragma unroll
  for (int e = 0; e < SOLVE_PREFETCH; ++e)
    if (e < E) acc = fma(pv[e], f[min(q + Q * e, w - 1)], acc);  // entries beyond column w - 1 are stored as zeros
  for (int e = SOLVE_PREFETCH; e < E; ++e)
    if (active) acc = fma(sp[(long long)e * TS], f[min(q + Q * e, w - 1)], acc);
  if (active) part[tid] = acc;
  __syncthreads();
  if (tid < ro) {
    double s2 = 0.0;
    for (int qq = 0; qq < Q; ++qq) s2 += part[qq * ro + tid];
    if (tid < top) {
      post_f64(xhat + T.c0 + tid, s2);  // X has a unit diagonal: f[tid] is inside the product
    } else {
      const int j = tid - top;  // update row a0 + j of the front, staged at f[w + j]
      post_f64(uvec + T.uoff + a0 + j, f[w + j] + s2);
    }
  }
}

__device__ __forceinline__ void dev_solve_bwd(const SolveItem& T, const double* __restrict__ SPb,
                                              const int* __restrict__ rows, double* __restrict__ y,
                                              double* __restrict__ xhat, double* __restrict__ uvec,
                                              double* __restrict__ ysol, double* __restrict__ spart, double* lds,
                                              int* __restrict__ info) {
  const int tid = threadIdx.x;
  const int w = T.w, P = T.Pb, E = T.Eb;
  const int a0 = T.a0, nu = T.a1 - T.a0;
  const int top = (T.sl == 0) ? w : 0;
  const int ro = top + nu;  // rows of this item's panel copy: [pivot rows (slice 0);] update rows [a0, a1)
  const int TS = w * P;
  const bool active = tid < TS;
  const int p = active ? tid / w : 0;
  double* tv = lds;         // ro: [x^; g]
  double* part = lds + ro;  // P * w <= 1024
  const double* __restrict__ sp = SPb + T.spb + tid;
  double pv[SOLVE_PREFETCH];
#pragma unroll
  for (int e = 0; e < SOLVE_PREFETCH; ++e) pv[e] = (active && e < E) ? sp[(long long)e * TS] : 0.0;
  const int myrow = (tid >= top && tid < ro) ? rows[T.rowoff + w + a0 + (tid - top)] : -1;
  // x^ from the forward item of this front (the root turns around here), the ancestors' solution
  // entries from their backward items: polled one by one, no flag, no fence
  if (tid < top) {
    tv[tid] = poll_f64(xhat + T.c0 + tid, info);
    sent_f64_agent(xhat + T.c0 + tid);  // single consumer: slot ready for the next solve
  } else if (myrow >= 0) {
    tv[tid] = poll_f64(ysol + myrow, info);
  }
  // the parent's forward item consumed this front's update vector long ago (it precedes the root's turn)
  if (tid >= top && tid < ro) sent_f64_agent(uvec + T.uoff + a0 + (tid - top));
  __syncthreads();
  double acc = 0.0;
#pragma unroll
  for (int e = 0; e < SOLVE_PREFETCH; ++e)
    if (e < E) acc = fma(pv[e], tv[min(p + P * e, ro - 1)], acc);  // entries beyond row ro - 1 are stored as zeros
  for (int e = SOLVE_PREFETCH; e < E; ++e)
    if (active) acc = fma(sp[(long long)e * TS], tv[min(p + P * e, ro - 1)], acc);
  if (active) part[tid] = acc;
  __syncthreads();
  if (tid < w) {
    double s2 = 0.0;
    for (int pp = 0; pp < P; ++pp) s2 += part[pp * w + tid];
    if (T.sl > 0) {
      post_f64(spart + T.poff + (long long)(T.sl - 1) * w + tid, s2);  // polled by slice 0 of the front
    } else {
      // the other slices' partial sums (they only waited for their ancestors' entries: usually there already),
      // added in slice order; the slots go back to the sentinel (single consumer)
      // (requested sixteen at a time: a loop of polls would pay one memory round trip per slice)
      for (int s0 = 1; s0 < T.nsl; s0 += 16) {
        unsigned long long bits[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int s = s0 + j;
          bits[j] = s < T.nsl ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(
                                                      spart + T.poff + (long long)(s - 1) * w + tid),
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int s = s0 + j;
          if (s < T.nsl) {
            double* slot = spart + T.poff + (long long)(s - 1) * w + tid;
            if (bits[j] == SOLVE_SENT) bits[j] = (unsigned long long)__double_as_longlong(poll_f64(slot, info));
            s2 += __longlong_as_double((long long)bits[j]);
            sent_f64_agent(slot);
          }
        }
      }
      y[T.c0 + tid] = s2;
      post_f64(ysol + T.c0 + tid, s2);
    }
  }
}

// The back substitution of the leaf columns, z_x = b~_x - A^^T y^ and z_y = D y^ (k_x_saddle), as the last
// workgroups of the fused solve launch: they poll the solution copy the backward items post (ysol) instead of
// waiting for a kernel boundary - the boundary behind the tree cost more than the 9 us of the product itself
// (two launches per steady-state solve: 104 us, of which 71 in the kernels).  X.z null: the product has its own launch.
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
__device__ __forceinline__ void dev_x_update(const XupdIn& X, int xb, const double* __restrict__ ys, int m,
                                             int* __restrict__ info, double* lds) {
  // Same lanes per column and the same summation order as k_x_saddle (XL = 8: identical bits).  These workgroups
  // become resident while the last backward items are still running, one per CU: everything that does not depend on
  // y - column pointers, values, row indices of XP columns per lane group, the scales of the y part - is requested
  // first, then all entries of y are requested in one batch and only those that still hold the sentinel are polled.
  constexpr int XL = 8, XE = 2, XP = 4;
  const int sub = threadIdx.x % XL;
  const int cpb = ST / XL;
  const int stride = X.nblocks * cpb;
  const SaddleMaps& M = X.M;
  const unsigned long long* __restrict__ yb = reinterpret_cast<const unsigned long long*>(ys);
  // the y part: z_y = D y^ (at most a few entries per thread)
  const int k0 = xb * ST + threadIdx.x, kstride = X.nblocks * ST;
  int yi = -1;
  double ysc = 0.0, dsum = 0.0;
  if (k0 < m) {
    yi = ext_row(M, X.perm[k0]);
    ysc = M.dscale[k0];
  }
  for (int jb = xb * cpb + threadIdx.x / XL; jb < X.n; jb += XP * stride) {
    double val[XP][XE];
    int idx[XP][XE], enext[XP], eend[XP];
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int j = jb + p * stride;
      const bool ok = j < X.n;
      const int e0 = ok ? X.Kp[j] + 1 + sub : 0, e1 = ok ? X.Kp[j + 1] : 0;
#pragma unroll
      for (int t = 0; t < XE; ++t) {
        const int e = e0 + XL * t;
        val[p][t] = (e < e1) ? X.Ksc[e] : 0.0;
        idx[p][t] = (e < e1) ? X.Kc_y[e] : -1;
      }
      enext[p] = e0 + XL * XE;
      eend[p] = e1;
    }
    unsigned long long bits[XP][XE];
#pragma unroll
    for (int p = 0; p < XP; ++p)
#pragma unroll
      for (int t = 0; t < XE; ++t)
        bits[p][t] = (idx[p][t] >= 0) ? __hip_atomic_load(yb + idx[p][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int j = jb + p * stride;
      double s = 0.0;
#pragma unroll
      for (int t = 0; t < XE; ++t)
        if (idx[p][t] >= 0) {
          const double yv = (bits[p][t] == SOLVE_SENT) ? poll_f64(ys + idx[p][t], info) : __longlong_as_double((long long)bits[p][t]);
          s += val[p][t] * yv;
        }
      for (int e = enext[p]; e < eend[p]; e += XL) s += X.Ksc[e] * poll_f64(ys + X.Kc_y[e], info);
#pragma unroll
      for (int o = XL / 2; o > 0; o >>= 1) s += __shfl_down(s, o, XL);
      if (sub == 0 && j < X.n) {
        const int v = M.vmap ? M.vmap[j] : -1;
        const double bj = X.b[j];
        if (v >= 0) {
          const double beta = X.b[v];
          const double mult = (bj - beta) - s;
          dsum += bj * beta;
          if (X.acc) {
            X.z[j] += beta;
            X.z[v] += mult;
          } else {
            X.z[j] = beta;
            X.z[v] = mult;
          }
        } else if (X.acc) {
          X.z[j] += bj - s;
        } else {
          const double zj = bj - s;
          dsum += bj * zj;
          X.z[j] = zj;
        }
      }
    }
  }
  if (X.dot_out) {  // (uniform; plain assignment mode only)
    dsum = wave_sum(dsum);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0;
      for (int q = 0; q < ST / 64; ++q) a += lds[q];
      X.dot_out[xb] = a;
    }
  }
  for (int k = k0; k < m; k += kstride) {
    const int i = (k == k0) ? yi : ext_row(M, X.perm[k]);
    if (i >= 0) {
      const double v = poll_f64(ys + k, info) * ((k == k0) ? ysc : M.dscale[k]);
      if (X.acc)
        X.z[i] += v;
      else
        X.z[i] = v;
    }
  }
}

__global__ __launch_bounds__(ST) void k_solve_tree(const SolveItem* __restrict__ items, int nf,
                                                   const double* __restrict__ SPf, const double* __restrict__ SPb,
                                                   const long long* __restrict__ xuoff,
                                                   const int* __restrict__ xinvoff, const int* __restrict__ inv,
                                                   const int* __restrict__ rows, double* __restrict__ y,
                                                   double* __restrict__ xhat,
                                                   double* __restrict__ uvec, double* __restrict__ ysol2, int m,
                                                   int* __restrict__ epoch, int* __restrict__ info,
                                                   const int* __restrict__ skip, RhsIn R, DecideIn D,
                                                   double* __restrict__ spart, XupdIn X) {
  __shared__ __attribute__((aligned(16))) double lds[2 * 1024 + 8];
  const int b = blockIdx.x;
  if (b == 2 * nf) {
    // one workgroup more than the tree has items: the verdict on the PREVIOUS solve, if its graph carried none
    // (steady state of a well-conditioned factorisation: no correction pass, and this saves the one-block launch
    // and its kernel boundary behind every solve; an entry point that needs the verdict earlier launches it)
    if (D.ctl && D.ctl->pending) dev_refine_decide(D, 1);
    return;
  }
  if (skip && *skip) return;
  const int par = *epoch & 1;  // constant while anybody reads it: advanced by the kernel behind this one, or by the
                               // LAST workgroup of this launch (workgroups are dispatched in index order, so every
                               // other one has read it by the time the last one is running)
  if (b > 2 * nf) {
    dev_x_update(X, b - 2 * nf - 1, ysol2 + (size_t)par * m, m, info, lds);
    if (b == (int)gridDim.x - 1) {
      __syncthreads();
      if (threadIdx.x == 0) *epoch += 1;
    }
    return;
  }
  if (b < nf) {
    const SolveItem& T = items[b];
    dev_solve_fwd(T, SPf, xuoff, xinvoff, inv, y, xhat, uvec, ysol2 + (size_t)(1 - par) * m, lds, info, R);
  } else {
    const SolveItem& T = items[b];  // second half of the list: the backward order
    dev_solve_bwd(T, SPb, rows, y, xhat, uvec, ysol2 + (size_t)par * m, spart, lds, info);
  }
}

// Solve panels of one front from its factored panel (X = inv(L11) strictly lower + pivots on the
// diagonal, L21 below): W = L21 X on the matrix cores, then both thread-major copies.  One workgroup
// (8 waves) per front, X in LDS; a wave owns 16-row strips of L21, requests its whole strip (the A
// operands of every step) in one batch, and keeps the strip of W (up to 8 tiles of 16 x 16) in its
// accumulators.  Padding entries of the arenas and the zeros above the diagonal of X are zero from
// the upload of the plan and never written.
constexpr int SPB = 512;
__device__ __forceinline__ void dev_build_solve_panel(const SolveItem& T, const double* __restrict__ L,
                                                      double* __restrict__ SPf, double* __restrict__ SPb, double* lds) {
  const int w = T.w, r = T.r;
  // the item's rows of S: the pivot rows (slice 0 only) and the update rows [ua, ua + u) of the front
  const int ua = T.a0, u = T.a1 - T.a0;
  const int top = (T.sl == 0) ? w : 0;
  const int ro = top + u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int nbk = (w + 15) >> 4, wp = nbk << 4;
  const int ldx = wp + 1;                    // odd leading dimension: column reads of X spread over the banks
  const double* __restrict__ Pn = L + T.Loff;
  double* X = lds;                           // wp x wp (ld ldx), unit lower, zero padded
  double* dinv = X + (size_t)ldx * wp;       // wp
  double* tile = dinv + wp + (size_t)wave * (16 * 17);  // per wave: one 16 x 16 tile, row stride 17
  // where column k of the forward copy and row i of the backward copy start (integer divisions are ~40
  // instructions each on this hardware: once per column / row instead of once per element)
  long long* offF = reinterpret_cast<long long*>(dinv + wp + (SPB / 64) * (16 * 17));  // wp
  long long* offB = offF + wp;                                                          // ro
  // the first strip of this wave: requested before X is staged, so that both arrive together
  const int nstrip = (u + 15) >> 4;
  double av[32];
  {
    const int a0 = wave << 4;
    const bool rowok = wave < nstrip && a0 + li < u;
    const double* __restrict__ Lr = Pn + w + ua + a0 + li;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const int j = 4 * t + lk;
      av[t] = (rowok && j < w) ? Lr[(long long)j * r] : 0.0;
    }
  }
  // X: eight independent loads per thread and batch (a load -> store loop would pay one memory round trip per
  // element)
  for (int k0 = wave; k0 < wp; k0 += 8 * (SPB / 64)) {  // eight columns per wave and batch, rows lane, lane + 64
    double v[8][2];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int k = k0 + t * (SPB / 64), i = lane + 64 * h2;
        v[t][h2] = (i == k) ? 1.0 : 0.0;
        if (i < w && k < w && i > k) v[t][h2] = Pn[i + (long long)k * r];
      }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int k = k0 + t * (SPB / 64), i = lane + 64 * h2;
        if (k < wp && i < wp) X[i + k * ldx] = v[t][h2];
      }
  }
  for (int k = tid; k < wp; k += SPB) dinv[k] = (k < w) ? 1.0 / Pn[k + (long long)k * r] : 1.0;
  const int Qf = T.Qf, Pb = T.Pb;
  const long long TSf = (long long)ro * Qf, TSb = (long long)w * Pb;
  for (int k = tid; k < wp; k += SPB) offF[k] = (long long)(k / Qf) * TSf + (long long)(k % Qf) * ro;
  for (int i = tid; i < ro; i += SPB) offB[i] = (long long)(i / Pb) * TSb + (long long)(i % Pb) * w;
  __syncthreads();
  double* __restrict__ sf = SPf + T.spf;
  double* __restrict__ sb = SPb + T.spb;
  // update rows: S[w + a, k] = -W[a, k], W = L21 X.  Strip of 16 rows per wave and turn.
  for (int st = wave; st < nstrip; st += SPB / 64) {
    const int a0 = st << 4;
    d4_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
    // columns j of L21 four at a time (A operand: rows a0 + li, columns j0 + lk); X[j, k] = 0 for j < k:
    // only the column tiles kt with 16 kt <= j0 + 3 take part
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const int j0 = 4 * t;
      if (j0 < wp) {
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
          if (16 * kt <= j0 + 3 && kt < nbk) acc[kt] = MFMA_F64(av[t], X[(j0 + lk) + (16 * kt + li) * ldx], acc[kt]);
      }
    }
    // the next strip of this wave travels while this one is stored
    {
      const int an = (st + SPB / 64) << 4;
      const bool rowok = st + SPB / 64 < nstrip && an + li < u;
      const double* __restrict__ Lr = Pn + w + ua + an + li;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        const int j = 4 * t + lk;
        av[t] = (rowok && j < w) ? Lr[(long long)j * r] : 0.0;
      }
    }
    // finished strip: tile by tile through the wave's LDS tile, row-major for the backward copy
    // (16 consecutive columns per store) and column-major for the forward copy (16 consecutive rows)
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt >= nbk) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = lk + 4 * q, k = 16 * kt + li, i = top + a0 + row;
        tile[row * 17 + li] = -acc[kt][q];
        if (a0 + row < u && k < w) sb[offB[i] + k] = -acc[kt][q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = li, col = lk + 4 * q, k = 16 * kt + col, i = top + a0 + row;
        if (a0 + row < u && k < w) sf[offF[k] + i] = tile[row * 17 + col];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // pivot rows: S[i, k] = X[i, k] (lower triangle); backward copy divided by d_i.  Two passes so that each
  // copy is written along its contiguous direction (rows i forward, columns k backward).
  if (top == 0) return;  // (the pivot rows belong to slice 0)
  for (int k = wave; k < w; k += SPB / 64)
    for (int i = k + lane; i < w; i += 64) sf[offF[k] + i] = X[i + k * ldx];
  for (int i = wave; i < w; i += SPB / 64) {
    const double di = dinv[i];
    for (int k = lane; k <= i; k += 64) sb[offB[i] + k] = X[i + k * ldx] * di;
  }
}
__global__ __launch_bounds__(SPB) void k_build_solve_panels(const SolveItem* __restrict__ items,
                                                            const double* __restrict__ L, double* __restrict__ SPf,
                                                            double* __restrict__ SPb) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  dev_build_solve_panel(items[blockIdx.x], L, SPf, SPb, lds);
}

// ---------------------------------------------------------------------------
// Saddle-point front / back end: K = [I A^T; A 0].
//
// Internally the constraint rows are EQUILIBRATED: A^ = D A with D = diag(2^-e_k), |row k of A^| in
// [0.7, 1.42) (powers of two: every scaling below is exact), S^ = A^ A^^T has a unit-order
// diagonal, y^ = D^-1 y.  MA57, the reference's indefinite backend, scales by default as well
// (fact_ma57.c:743 keeps ICNTL(15) of ma57id_).  With unit rows cond(S^) and cond(K^) agree up to
// ||A^||^2 <= max row overlap, so the x-before-y pivot order loses nothing against a pivoted
// factorisation of K^ once the solve is refined on K itself.
//
//   t_p  = A^_p b~_x - D b_y[perm]      (right-hand side of S^ y^ = t)
//   z_x  = b~_x - A^^T y^ ;  z_y = D y^  (back substitution of the leaf columns)
//   res  = b - K z                       (iterative refinement; norms in the equilibrated space)
// A^_p is the CSR of A^ with rows in pivot order; the x update walks the columns
// of K itself (its CSC arrays are the CSR of A^T).
//
// Working-set maps (device assembly with a cached superset plan, hipfact_assemble_kkt): the
// factorised structure covers the constraint rows of a SUPERSET of the working set and never
// contains the unit rows of active bounds.  SaddleMaps translates between the caller's vectors
// (length n + |W|, working-set numbering) and that structure:
//   cmap[s]  position of structure row s in the caller's vectors, -1: row not in the working set
//            (its values are zero, its pivot is 1, its multiplier 0)
//   vmap[j]  position of the unit row of the active bound of x_j, -1: bound inactive.
// An active bound x_j = beta is eliminated exactly: b~_j = beta replaces b_j in the products, the
// Schur complement is formed without column j (Kprod), and afterwards x_j = beta,
// y_bound = b_j - beta - (A^T y)_j.  Both maps null: the structure IS the caller's K.
// ---------------------------------------------------------------------------

// Device-side control block of the iterative refinement (no host round trip per solve): written
// by the last block of every residual kernel, read by the kernels of the correction passes,
// which return at once when `done` is set.

// 16 lanes per row (rows of A hold ~20 entries in the headline configuration):
// consecutive lanes read consecutive entries, fixed shuffle tree => deterministic.

// Row equilibration, values of A^ in pivot order (Ar_val), and the scaled copy of K's values in K's
// own order: Ksc for the x update, Kprod for the Schur-complement products (the same array unless
// there are active bounds, whose columns do not enter the products).  Every off-diagonal entry of
// a column < n of K belongs to exactly one row of A, so the scatter through Ar_src covers them
// all (the unit diagonal is never read).  norm^2 is taken over the columns that enter the products.
// The first nbz blocks carry the zero fill of the factor arena (and of the info words) along: this
// kernel is bound by dependent gathers and leaves the memory system idle, the fill is pure
// bandwidth, and as two graph nodes they would run one after the other on the critical path.
__global__ __launch_bounds__(FB) void k_row_scale(int m, const int* __restrict__ Ar_ptr,
                                                  const int* __restrict__ Ar_col, const int* __restrict__ Ar_src,
                                                  const double* __restrict__ Kval, const int* __restrict__ vmap,
                                                  const int* __restrict__ dmask, int enable, double* __restrict__ dscale,
                                                  double* __restrict__ Ar_val, double* __restrict__ Ar_full,
                                                  double* __restrict__ Ksc, double* __restrict__ Kprod, int nbz,
                                                  double2* __restrict__ zero,
                                                  long long nzero, int* __restrict__ info) {
  if ((int)blockIdx.x < nbz) {
    if (blockIdx.x == 0 && threadIdx.x < INFO_BYTES / 4) info[threadIdx.x] = 0;
    const double2 z = {0.0, 0.0};
    for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < nzero; i += (long long)nbz * FB) zero[i] = z;
    return;
  }
  const int sub = threadIdx.x % RL;
  const int rpb = FB / RL;
  const int nbw = gridDim.x - nbz, bid = blockIdx.x - nbz;
  const int iters = (m + nbw * rpb - 1) / (nbw * rpb);
  for (int it = 0; it < iters; ++it) {  // uniform trip count (the shuffles need whole groups)
    const int k = (it * nbw + bid) * rpb + threadIdx.x / RL;
    double s = 0.0;
    int p0 = 0, p1 = 0;
    constexpr int KEEP = 4;  // rows of up to 64 entries are read once
    double v[KEEP];
    int src[KEEP];
    bool fx[KEEP];
    if (k < m) {
      p0 = Ar_ptr[k];
      p1 = Ar_ptr[k + 1];
      int q = 0;
      for (int p = p0 + sub; p < p1; p += RL, ++q) {
        const int e = Ar_src[p];
        const double val = Kval[e];
        const bool fixed = vmap && vmap[Ar_col[p]] >= 0;
        // a dense column (dense_cols.inc) is masked out of every product of the engine - unless its bound is active
        const bool dense = dmask && !fixed && dmask[Ar_col[p]] >= 0;
        if (q < KEEP) {
          v[q] = val;
          src[q] = e;
          fx[q] = fixed;
        }
        if (!fixed && !dense) s += val * val;
      }
    }
#pragma unroll
    for (int o = RL / 2; o > 0; o >>= 1) s += __shfl_down(s, o, RL);
    s = __shfl(s, 0, RL);
    double d = 1.0;
    if (enable && s > 0.0 && s < 1.7e308) {
      int e;
      (void)frexp(s, &e);         // s = f 2^e, f in [0.5, 1)
      d = ldexp(1.0, -(e >> 1));  // d^2 s in [0.5, 2)
    }
    if (k < m) {
      if (sub == 0) dscale[k] = d;
      int q = 0;
      for (int p = p0 + sub; p < p1; p += RL, ++q) {
        int e;
        double val;
        bool fixed;
        if (q < KEEP) {
          e = src[q];
          val = v[q];
          fixed = fx[q];
        } else {
          e = Ar_src[p];
          val = Kval[e];
          fixed = vmap && vmap[Ar_col[p]] >= 0;
        }
        val *= d;
        if (Ar_full) {
          Ar_full[p] = val;  // the residual is taken on K itself
          if (!fixed && dmask[Ar_col[p]] >= 0) val = 0.0;
        }
        Ar_val[p] = val;
        Ksc[e] = val;
        if (Kprod != Ksc) Kprod[e] = fixed ? 0.0 : val;
      }
    }
  }
}

// 8 lanes per column of K (columns hold ~11 entries)
constexpr int CL = 8;

// rows of the superset that are not in the working set: unit pivot (their row of A is zero)
__global__ __launch_bounds__(FB) void k_diag_inactive(int m, const int* __restrict__ perm,
                                                      const int* __restrict__ cmap,
                                                      const long long* __restrict__ diag_target,
                                                      double* __restrict__ L) {
  for (int k = blockIdx.x * FB + threadIdx.x; k < m; k += gridDim.x * FB)
    if (cmap[perm[k]] < 0) L[diag_target[k]] = 1.0;
}

__global__ __launch_bounds__(FB) void k_rhs_saddle(int m, const int* __restrict__ Ar_ptr,
                                                   const int* __restrict__ Ar_col, const double* __restrict__ Ar_val,
                                                   const int* __restrict__ perm, SaddleMaps M,
                                                   const double* __restrict__ b, double* __restrict__ t,
                                                   const int* __restrict__ skip) {
  if (skip && *skip) return;
  const RhsIn R{Ar_ptr, Ar_col, Ar_val, perm, M, b};
  const int sub = threadIdx.x % RL;
  const int rpb = FB / RL;
  for (int k = blockIdx.x * rpb + threadIdx.x / RL; k < m; k += gridDim.x * rpb) {
    const double s = rhs_row(R, k, sub);
    if (sub == 0) t[k] = s;
  }
}

// ACC: z += (correction pass) instead of z =
template <bool ACC>
__global__ __launch_bounds__(FB) void k_x_saddle(int n, int m, const int* __restrict__ Kp,
                                                 const double* __restrict__ Ksc, const int* __restrict__ Kc_y,
                                                 const int* __restrict__ perm, SaddleMaps M,
                                                 const double* __restrict__ yp, const double* __restrict__ b,
                                                 double* __restrict__ z, const int* __restrict__ skip,
                                                 int* __restrict__ epoch) {
  if (skip && *skip) return;
  if (epoch && blockIdx.x == 0 && threadIdx.x == 0) *epoch += 1;  // the fused solve launch before this one is over
  const int sub = threadIdx.x % CL;
  const int cpb = FB / CL;
  for (int j = blockIdx.x * cpb + threadIdx.x / CL; j < n; j += gridDim.x * cpb) {
    double s = 0.0;
    const int e1 = Kp[j + 1];
    for (int e = Kp[j] + 1 + sub; e < e1; e += CL) s += Ksc[e] * yp[Kc_y[e]];
#pragma unroll
    for (int o = CL / 2; o > 0; o >>= 1) s += __shfl_down(s, o, CL);
    if (sub == 0) {
      const int v = M.vmap ? M.vmap[j] : -1;
      if (v >= 0) {
        const double beta = b[v];
        const double mult = (b[j] - beta) - s;
        if (ACC) {
          z[j] += beta;
          z[v] += mult;
        } else {
          z[j] = beta;
          z[v] = mult;
        }
      } else if (ACC) {
        z[j] += b[j] - s;
      } else {
        z[j] = b[j] - s;
      }
    }
  }
  for (int k = blockIdx.x * FB + threadIdx.x; k < m; k += gridDim.x * FB) {
    const int i = ext_row(M, perm[k]);
    if (i >= 0) {
      const double v = yp[k] * M.dscale[k];
      if (ACC)
        z[i] += v;
      else
        z[i] = v;
    }
  }
}

// ---- refinement control ---------------------------------------------------------------------
// Every block of a residual kernel leaves (max |r^|, max |b^|, max |z^|); the block that arrives
// last reduces them and decides.  A NaN residual propagates (it must end the loop).
//   tol = clamp(target / kappa, 4.5e-16, 1e-12), kappa = 10 max|d| / min|d| of the pivots: the
//   forward error of a solve with backward error omega is about kappa omega, so well conditioned
//   systems are accepted after the first pass and ill conditioned ones are refined to the limit.

// block partials of a residual kernel: plain stores, the decision is taken by the kernel behind it
__device__ __forceinline__ void refine_partials(double* __restrict__ partials, double mr, double mb, double mz) {
  __shared__ double sh[3][FB / 64];
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mr = nanmax(mr, __shfl_down(mr, o, 64));
    mb = nanmax(mb, __shfl_down(mb, o, 64));
    mz = nanmax(mz, __shfl_down(mz, o, 64));
  }
  if ((tid & 63) == 0) {
    sh[0][tid >> 6] = mr;
    sh[1][tid >> 6] = mb;
    sh[2][tid >> 6] = mz;
  }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < FB / 64; ++q) {
      mr = nanmax(mr, sh[0][q]);
      mb = nanmax(mb, sh[1][q]);
      mz = nanmax(mz, sh[2][q]);
    }
    partials[3 * blockIdx.x] = mr;
    partials[3 * blockIdx.x + 1] = mb;
    partials[3 * blockIdx.x + 2] = mz;
  }
}

// res = b - K z for the saddle matrix (columns < n of the lower CSC hold I and A), in the caller's
// numbering.  Two sweeps with cooperative lanes (8 per column of K, 16 per row of A); the last
// block updates the refinement control block (ctl null: plain residual).
__global__ __launch_bounds__(FB) void k_residual_saddle(int n, int m, const int* __restrict__ Kp,
                                                        const int* __restrict__ Ki, const double* __restrict__ Kval,
                                                        const int* __restrict__ Ar_ptr, const int* __restrict__ Ar_col,
                                                        const double* __restrict__ Ar_val,
                                                        const int* __restrict__ perm, SaddleMaps M,
                                                        const double* __restrict__ b, const double* __restrict__ z,
                                                        double* __restrict__ res, const RefineCtl* __restrict__ ctl,
                                                        double* __restrict__ partials, int first, int* __restrict__ defer) {
  if (ctl && !first && ctl->done) return;
  double mr = 0.0, mb = 0.0, mz = 0.0;
  // The two sweeps are chains of three dependent gathers each (pointer -> index / value -> vector entry);
  // the first half of the grid takes the columns of K, the second half the rows of A, side by side.
  const int nbx = (int)gridDim.x >> 1;
  if ((int)blockIdx.x < nbx) {
    const int sub = threadIdx.x % CL;
    const int cpb = FB / CL;
    const int iters = (n + nbx * cpb - 1) / (nbx * cpb);
    for (int it = 0; it < iters; ++it) {  // uniform trip count (the shuffles need whole groups)
      const int j = (it * nbx + blockIdx.x) * cpb + threadIdx.x / CL;
      double s = 0.0;
      if (j < n) {
        const int e1 = Kp[j + 1];
        for (int e = Kp[j] + sub; e < e1; e += CL) {
          const int i = Ki[e];
          const int zi = i < n ? i : ext_row(M, i - n);
          if (zi >= 0) s += Kval[e] * z[zi];
        }
      }
#pragma unroll
      for (int o = CL / 2; o > 0; o >>= 1) s += __shfl_down(s, o, CL);
      if (sub == 0 && j < n) {
        const double bj = b[j], zj = z[j];
        const int v = M.vmap ? M.vmap[j] : -1;
        if (v >= 0) {  // unit row of the active bound: multiplier z[v] in row j, and its own row x_j = b_v
          const double zv = z[v], bv = b[v];
          s += zv;
          const double rv = bv - zj;
          res[v] = rv;
          mr = nanmax(mr, fabs(rv));
          mb = fmax(mb, fabs(bv));
          mz = fmax(mz, fabs(zv));
        }
        const double rj = bj - s;
        res[j] = rj;
        mr = nanmax(mr, fabs(rj));
        mb = fmax(mb, fabs(bj));
        mz = fmax(mz, fabs(zj));
      }
    }
  } else {
    const int nby = (int)gridDim.x - nbx, by = (int)blockIdx.x - nbx;
    const int sub = threadIdx.x % RL;
    const int rpb = FB / RL;
    const int iters = (m + nby * rpb - 1) / (nby * rpb);
    for (int it = 0; it < iters; ++it) {
      const int k = (it * nby + by) * rpb + threadIdx.x / RL;
      double s = 0.0;
      if (k < m) {
        const int p1 = Ar_ptr[k + 1];
        for (int p = Ar_ptr[k] + sub; p < p1; p += RL) s += Ar_val[p] * z[Ar_col[p]];
      }
#pragma unroll
      for (int o = RL / 2; o > 0; o >>= 1) s += __shfl_down(s, o, RL);
      if (sub == 0 && k < m) {
        const int i = ext_row(M, perm[k]);
        if (i >= 0) {
          const double d = M.dscale[k];
          const double bi = b[i] * d;  // equilibrated row
          const double ri = bi - s;
          res[i] = ri / d;
          mr = nanmax(mr, fabs(ri));
          mb = fmax(mb, fabs(bi));
          mz = fmax(mz, fabs(z[i] / d));
        }
      }
    }
  }
  if (ctl) refine_partials(partials, mr, mb, mz);
  if (defer && blockIdx.x == 0 && threadIdx.x == 0) *defer = 1;  // picked up a kernel boundary (or more) later
}

// Generic mode residual: res = b - (L + L^T - diag) z with L lower CSC and its
// transpose (CSR of L) both resident.
__global__ __launch_bounds__(FB) void k_residual_sym(int N, const int* __restrict__ Kp, const int* __restrict__ Ki,
                                                     const double* __restrict__ Kval, const int* __restrict__ Tp,
                                                     const int* __restrict__ Ti, const int* __restrict__ Tsrc,
                                                     const double* __restrict__ b, const double* __restrict__ z,
                                                     double* __restrict__ res, const RefineCtl* __restrict__ ctl,
                                                     double* __restrict__ partials, int first, int* __restrict__ defer) {
  if (ctl && !first && ctl->done) return;
  double mr = 0.0, mb = 0.0, mz = 0.0;
  const int iters = (N + gridDim.x * FB - 1) / (gridDim.x * FB);
  for (int it = 0; it < iters; ++it) {
    const int j = (it * gridDim.x + blockIdx.x) * FB + threadIdx.x;
    if (j < N) {
      const double bj = b[j];
      double s = bj;
      for (int e = Kp[j]; e < Kp[j + 1]; ++e) s -= Kval[e] * z[Ki[e]];  // column j: rows >= j
      for (int p = Tp[j]; p < Tp[j + 1]; ++p)                            // row j: columns < j
        if (Ti[p] != j) s -= Kval[Tsrc[p]] * z[Ti[p]];
      res[j] = s;
      mr = nanmax(mr, fabs(s));
      mb = fmax(mb, fabs(bj));
      mz = fmax(mz, fabs(z[j]));
    }
  }
  if (ctl) refine_partials(partials, mr, mb, mz);
  if (defer && blockIdx.x == 0 && threadIdx.x == 0) *defer = 1;  // picked up a kernel boundary (or more) later
}

// y += a x, or nothing when *skip is set
__global__ __launch_bounds__(FB) void k_axpy(long long n, double a, const double* __restrict__ x,
                                             double* __restrict__ y, const int* __restrict__ skip) {
  if (skip && *skip) return;
  for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < n; i += (long long)gridDim.x * FB)
    y[i] += a * x[i];
}

// ---------------------------------------------------------------------------
// Vector kernels of the device-resident projected CG (tr/steihaug_solver.c).
// Dot products leave one partial per block (fixed grid, fixed order: the host
// sums them in index order => deterministic).
// ---------------------------------------------------------------------------
constexpr int DOT_BLOCKS = 128;

// out[3 * blockIdx + t] = partial of <x_t, y_t>, t = 0, 1, 2 (null pointers skip a pair)
__global__ __launch_bounds__(FB) void k_dots3(int n, const double* __restrict__ x0, const double* __restrict__ y0,
                                              const double* __restrict__ x1, const double* __restrict__ y1,
                                              const double* __restrict__ x2, const double* __restrict__ y2,
                                              double* __restrict__ out) {
  __shared__ double sh[3][FB / 64];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = blockIdx.x * FB + threadIdx.x; i < n; i += gridDim.x * FB) {
    if (x0) s0 += x0[i] * y0[i];
    if (x1) s1 += x1[i] * y1[i];
    if (x2) s2 += x2[i] * y2[i];
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s0;
    sh[1][threadIdx.x >> 6] = s1;
    sh[2][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int q = 0; q < FB / 64; ++q) s += sh[threadIdx.x][q];
    out[3 * blockIdx.x + threadIdx.x] = s;
  }
}

// y = a x
__global__ __launch_bounds__(FB) void k_scale_to(int n, double a, const double* __restrict__ x, double* __restrict__ y) {
  for (int i = blockIdx.x * FB + threadIdx.x; i < n; i += gridDim.x * FB) y[i] = a * x[i];
}

// s = Q c: Q is n x k column-major (the Lanczos basis kept in HBM), c on the device; fixed summation order
__global__ __launch_bounds__(FB) void k_combine(int n, int k, const double* __restrict__ Q,
                                                const double* __restrict__ c, double* __restrict__ s) {
  for (int i = blockIdx.x * FB + threadIdx.x; i < n; i += gridDim.x * FB) {
    double acc = 0.0;
    for (int j = 0; j < k; ++j) acc += c[j] * Q[i + (long long)j * n];
    s[i] = acc;
  }
}

// y = a x + b y
__global__ __launch_bounds__(FB) void k_axpby(int n, double a, const double* __restrict__ x, double b,
                                              double* __restrict__ y) {
  for (int i = blockIdx.x * FB + threadIdx.x; i < n; i += gridDim.x * FB) y[i] = a * x[i] + b * y[i];
}

__global__ __launch_bounds__(FB) void k_scatter(long long n, const int* __restrict__ idx,
                                                const double* __restrict__ in, double* __restrict__ out,
                                                int* __restrict__ epoch) {
  if (epoch && blockIdx.x == 0 && threadIdx.x == 0) *epoch += 1;
  for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < n; i += (long long)gridDim.x * FB)
    out[idx[i]] = in[i];
}

// generic mode, correction passes: gather / scatter-accumulate unless *skip is set
__global__ __launch_bounds__(FB) void k_gather_skip(long long n, const int* __restrict__ src,
                                                    const double* __restrict__ in, double* __restrict__ out,
                                                    const int* __restrict__ skip) {
  if (skip && *skip) return;
  for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < n; i += (long long)gridDim.x * FB)
    out[i] = in[src[i]];
}
__global__ __launch_bounds__(FB) void k_scatter_acc(long long n, const int* __restrict__ idx,
                                                    const double* __restrict__ in, double* __restrict__ out,
                                                    const int* __restrict__ skip, int* __restrict__ epoch) {
  if (skip && *skip) return;
  if (epoch && blockIdx.x == 0 && threadIdx.x == 0) *epoch += 1;
  for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < n; i += (long long)gridDim.x * FB)
    out[idx[i]] += in[i];
}

// sparse right-hand side -> dense (sleqp_vec_to_raw, sparse/vec.c:105-119); out pre-zeroed
__global__ __launch_bounds__(FB) void k_scatter_sparse(int nnz, const int* __restrict__ idx,
                                                       const double* __restrict__ val, double* __restrict__ out) {
  for (int i = blockIdx.x * FB + threadIdx.x; i < nnz; i += gridDim.x * FB) out[idx[i]] = val[i];
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

// Values of the structure K_s = [I J_s^T; J_s 0] of a superset plan (same column order as
// fill_aug_jac without the bound rows): rows of the superset that are not in the working set get
// zeros, so that the plan of the superset serves every working set inside it without re-analysis.
// Also refreshes the maps between the structure and the caller's numbering.
__global__ __launch_bounds__(FB) void k_struct_fill(int n, int ms, const int* __restrict__ jp,
                                                    const int* __restrict__ ji, const double* __restrict__ jx,
                                                    const int* __restrict__ var_index,
                                                    const int* __restrict__ cons_index,
                                                    const int* __restrict__ sidx, const int* __restrict__ srow,
                                                    const int* __restrict__ kp, double* __restrict__ kx,
                                                    int* __restrict__ vmap, int* __restrict__ cmap) {
  for (int j = blockIdx.x * FB + threadIdx.x; j < n; j += gridDim.x * FB) {
    int e = kp[j];
    kx[e++] = 1.0;
    for (int q = jp[j]; q < jp[j + 1]; ++q) {
      const int i = ji[q];
      if (sidx[i] >= 0) kx[e++] = cons_index[i] >= 0 ? jx[q] : 0.0;
    }
    const int vi = var_index[j];
    vmap[j] = vi >= 0 ? n + vi : -1;
  }
  for (int s = blockIdx.x * FB + threadIdx.x; s < ms; s += gridDim.x * FB) {
    const int ci = cons_index[srow[s]];
    cmap[s] = ci >= 0 ? n + ci : -1;
  }
}

}  // namespace hipfact
