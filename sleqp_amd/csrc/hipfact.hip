// Host runtime and C ABI (include/hipfact.h) of the hipfact KKT backend.
//
// One handle = one backend instance = one HIP device + one stream, like one
// SleqpFact object in the reference (fact/fact.c:21-45); no process-global
// state.  All numerics run on the device; there is no CPU fallback: without a
// usable GPU hipfact_create fails with HIPFACT_EDEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/hipfact.h"
#include "device_types.h"
#include "plan.h"
#include "tridiag_tr.h"

// single translation unit: the kernels are compiled together with their launcher
#include "kernels.hip"
#define DENSE_COLS_KERNELS
#include "dense_cols.inc"
#undef DENSE_COLS_KERNELS
#define KRYLOV_DEVICE_KERNELS
#include "krylov_device.inc"
#undef KRYLOV_DEVICE_KERNELS

namespace hipfact {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
    o.p = nullptr;
    o.bytes = 0;
  }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      bytes = o.bytes;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    release();
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

struct PinBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~PinBuf() {
    if (p) (void)hipHostFree(p);
  }
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

struct LevelInfo {
  int begin = 0, count = 0;
  size_t lds_factor = 0, lds_fwd = 0, lds_bwd = 0;
  // split mode (few, large fronts): one kernel per phase, many workgroups per front
  bool split = false;
  bool pull = false;  // split kernels gather the children's updates themselves (no phase A launch)
  bool chain = false; // some front of the level has more than MAXCH children (descriptor chains)
  int nparts = 1;
  long long itA = 0;                    // offset (in ints) into d_items
  long long itB = 0, itC = 0, itD = 0;  // offsets (in FrontItems) into d_fitems
  int nA = 0, nC = 0, nD = 0;           // number of (front, part) items
  int panel_threads = 512;              // 16 panel rows per wave
  size_t lds_pivot = 0, lds_panel = 0, lds_schur = 0, lds_asm = 0, lds_solve_max = 0;
  // single-front level of a dense chain below the dataflow launch: its pivot and panel items as one small dataflow
  // launch (the panel workgroups follow the posted pivot block) instead of two launches
  long long mini_off = 0;  // offset (in TopFItems) into d_tfitems
  int mini_cnt = 0;
  size_t mini_lds = 0;
};

// kernel classes for the event-timed profiling mode (option "profile")
enum ProfClass { PC_MEMSET = 0, PC_MVALS, PC_GATHER, PC_FACTOR, PC_FACTOR_A, PC_FACTOR_B, PC_FACTOR_C, PC_FACTOR_D, PC_FACTOR_T, PC_FWD, PC_BWD, PC_RHS, PC_XUPD, PC_RESID, PC_AXPY, PC_PERM, PC_SPANEL, PC_TREE, PC_COUNT };
static const char* const kProfNames[PC_COUNT] = {"memset", "mvals", "gather", "factor", "factorA", "factorB", "factorC",
                                                 "factorD", "factorT", "fwd", "bwd", "rhs", "xupd", "resid", "axpy", "perm",
                                                 "spanel", "tree"};

struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;  // pairs (start, stop)
  std::vector<int> cls;
  size_t used = 0;
  double ms[PC_COUNT] = {0};
  long cnt[PC_COUNT] = {0};
  ~Prof() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};

}  // namespace hipfact

using namespace hipfact;

static thread_local std::string g_create_error;

struct GraphEntry {
  int kind;  // 0 factor, 1 solve (first pass, residual, in-graph correction passes), 2 continuation passes
  const void* b;
  void* z;
  int passes;  // correction passes inside the graph
  hipGraphExec_t exec;
};

// captured graphs of one plan state; owns the executables
struct GraphList {
  std::vector<GraphEntry> v;
  GraphList() = default;
  GraphList(const GraphList&) = delete;
  GraphList& operator=(const GraphList&) = delete;
  GraphList(GraphList&& o) noexcept : v(std::move(o.v)) { o.v.clear(); }
  GraphList& operator=(GraphList&& o) noexcept {
    if (this != &o) {
      clear();
      v = std::move(o.v);
      o.v.clear();
    }
    return *this;
  }
  ~GraphList() { clear(); }
  void clear() {
    for (auto& g : v) (void)hipGraphExecDestroy(g.exec);
    v.clear();
  }
  size_t size() const { return v.size(); }
  std::vector<GraphEntry>::iterator begin() { return v.begin(); }
  std::vector<GraphEntry>::iterator end() { return v.end(); }
  void push_back(const GraphEntry& g) { v.push_back(g); }
};

// Everything that belongs to ONE symbolic plan: the plan, its device image, the numeric arenas
// and the captured graphs (which hold pointers into exactly these buffers).  The handle IS the
// active state (it derives from this struct, so h->d_L etc. address the active plan); states of
// other sparsity patterns / working-set supersets wait in an LRU list and are swapped in whole,
// so coming back to a pattern seen before costs neither an analysis nor an upload nor a capture.
struct PlanState {
  Plan plan;
  bool have_plan = false, factored = false, solved = false;
  bool factor_checked = false;  // info words of the last factorisation have been read back
  bool prod_packed = false;     // product lists as one packed word per pair
  bool idx32 = false;           // product-list pointers and panel targets fit 32 bits
  unsigned long long key_hash = 0;  // FNV-1a of the pattern the plan was built for
  // superset plans (hipfact_assemble_kkt): pattern of J and the constraint rows the structure covers
  bool from_jacobian = false;
  std::vector<int> Jp, Ji;    // cons_jac pattern (CSC) the plan was built for
  std::vector<int> sidx;      // per constraint row of J: its row in the structure, -1 = not covered
  int m_struct = 0;           // rows covered
  bool maps_on = false;       // d_vmap / d_cmap translate between the caller's numbering and the structure
  int N_ext = 0;              // dimension of the caller's vectors (n + |W|); == plan.N without maps
  int n_bounds = 0;           // active bounds of the current working set
  unsigned long long use_stamp = 0;  // LRU clock
  int refine_inline = 1;      // correction passes currently carried by the solve graphs
  int seq_at_factor = 0;      // handle's solve_seq at the time of the last factorisation
  bool inline_probe = true;   // the first solve of this factorisation has not been looked at yet
  bool wc_hint = false;       // the previous factorisation of this plan was judged well-conditioned (first pass enough)
  GraphList graphs;
  int ftop_level = 1 << 30, ftop_count = 0;
  size_t ftop_lds = 0;
  // The solve-panel items of the dataflow launch leave the panels of the fronts BELOW the launch zeroed behind them
  // (level-major arena: a prefix of l_prefix doubles), so the fill in front of the next factorisation skips them.
  long long l_prefix = 0;  // 0: off for this plan
  bool L_clean = false;    // the prefix is zero right now
  long long mini_x_off = 0;  // posted pivot blocks of the chain levels' small dataflow launches in d_xarena (doubles) ...
  size_t mini_x_bytes = 0;   // ... and how many bytes: back to the sentinel with every factorisation
  double ent_fused = 0, ent_split = 0, rows_fused = 0, rows_split = 0;  // L entries / row indices per kernel family
  std::vector<LevelInfo> levels;
  // top of the tree solved in one launch per direction (levels >= top_level)
  int top_level = 1 << 30, top_count = 0;
  size_t top_lds_fwd = 0, top_lds_bwd = 0;
  // plan on device
  DevBuf d_sn, d_level_sn, d_rows, d_rel, d_child, d_Mtarget, d_prod_ptr, d_prod_a, d_prod_b, d_src;
  DevBuf d_items, d_fitems, d_top_sn, d_titems, d_flags, d_inv, d_tfitems, d_ftarget, d_wpart, d_pullx;
  DevBuf d_perm, d_Ar_ptr, d_Ar_col, d_Ar_src, d_Ar_val, d_Kp, d_Ki, d_Kc_y, d_Tp, d_Ti, d_Tsrc;
  // numeric
  DevBuf d_xarena;  // posted pivot blocks of the single-launch factorisation (polled by its panel workgroups)
  DevBuf d_ysol;    // polled copy of the solution of M y = t (single-launch backward sweep)
  DevBuf d_Kval, d_L, d_U, d_uvec, d_y, d_rhs, d_sol, d_res;
  DevBuf d_Ksc, d_Kprod, d_dscale, d_vmap, d_cmap, d_diag_target, d_sidx, d_srow;
  // fused solve (k_solve_tree): solve panels S = [X; -W] in two thread-major copies, x^ exchange slots
  bool fused_solve = false;
  size_t sp_lds = 0;
  double sp_bytes = 0;
  DevBuf d_SPf, d_SPb, d_sitems, d_xhat, d_sxuoff, d_sxinvoff, d_epoch, d_spart;
  int n_sitems = 0;  // items of the fused solve launch: one per front, row slices for fronts of more than 1024 rows
  // dense columns of A (dense_cols.inc): left out of S, applied to every solve by a rank-2k correction
  int nd = 0, dense_cap = 0;
  DevBuf d_dmask, d_dcols, d_Bq, d_Zq, d_dtmp, d_dG, d_dMinv, d_dw, d_dfix;
  DevBuf d_Ar_full;  // scaled values of A in pivot order WITH the dense columns (the residual is taken on K itself)

  PlanState() = default;
  PlanState(PlanState&&) = default;
  PlanState& operator=(PlanState&&) = default;
};

#define VTABLE_SUPERSET_TYPES
#include "vtable_superset.inc"
#undef VTABLE_SUPERSET_TYPES

struct hipfact_handle : PlanState {
  std::atomic<int> refcount{1};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;          // solve panels of the finished bottom levels are built beside the top-of-tree launch
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::string error;
  PlanParams prm;
  std::vector<std::unique_ptr<PlanState>> cache;  // inactive plan states, at most plan_cache_max
  int plan_cache_max = 4;
  bool spanel_side = false;       // solve panels of the bottom levels on a second stream beside k_factor_top (measured: no gain, the
                                  // latency-bound top-of-tree launch slows down by as much as the overlap saves: 0.949 vs 0.940 ms)
  bool chain_fuse = true;         // single-front levels of a dense chain: pivot + panel items as one small dataflow launch
  bool solve_slices = true;       // fused solve: fronts whose panel share does not fit the registers of one item are row-sliced
  bool speculate = true;          // set_matrix: queue values + factorisation before the pattern comparison has finished
  bool xupd_fused = true;       // ... and its last workgroups do the back substitution of the leaf columns (one launch per solve; 256 workgroups that request everything independent of y first and poll y in one batch: -4 us per solve against the separate 9 us launch)
  bool rhs_fused = true;          // fused solve: the forward items form their rows of the right-hand side themselves
  bool spanel_fold = true;        // solve panels as filler items of k_factor_top (else a launch of their own behind it)
  int spanel_fold_room = 224;     // ... as many per level as fit this many workgroup slots together with its pivot and panel items
  bool sp_folded = false;         // (result of the plan upload)
  bool solve_fused = true;        // one launch for the whole solve tree on the solve panels (when every front qualifies)
  bool assemble_superset = true;  // hipfact_assemble_kkt analyses a superset structure of J instead of K itself
  bool superset_vtable = true;    // hipfact_set_matrix recognises the rows of an augmented K and reuses superset plans (vtable_superset.inc)
  std::unique_ptr<VirtualJ> vj{new VirtualJ};
  DevBuf d_kin, d_vsrc;           // the caller's values of K; value map into the virtual Jacobian
  DevBuf d_retry_b;               // right-hand side kept across a retry on the exact row set
  long vtable_retries = 0;
  bool jdev_valid = false;        // pattern of the Jacobian resident in d_jp / d_ji
  unsigned long long jdev_hash = 0;
  int jdev_n = 0, jdev_nnz = 0;
  unsigned long long use_clock = 0;
  long plan_swaps = 0;
  // Iterative refinement on K itself, controlled on the device (RefineCtl): every solve graph holds
  // the first pass, the residual, and `refine_inline` correction passes whose kernels return at once
  // when the control block says "done".  Entry points that synchronise anyway (hipfact_solution,
  // hipfact_check, the dot products of the projected CG) continue a solve that is still not done,
  // up to `refine_max` passes in total, and remember how many it took for the next solves.
  int refine_steps = 1;          // correction passes inside the solve graph (0: no residual at all)
  int refine_max = 10;           // total passes including the host-continued ones
  bool refine_adaptive = true;   // false: every in-graph pass runs unconditionally
  double refine_tol = 1e-10;     // forward-error target: backward-error tolerance = refine_tol / kappa_est
  double fail_omega = 1e-8;      // a solve that stalls above this backward error is reported as singular
  bool equilibrate = true;       // row equilibration of the constraint block (saddle mode)
  long num_refined = 0;          // solves that applied at least one correction pass
  long num_passes = 0;           // correction passes applied in total
  bool decide_lazy = true;       // verdict of a solve without correction passes delivered by the next tree launch
  // Once a factorisation has been judged well-conditioned (its solves meet the tolerance in the first pass with room
  // to spare: no correction pass in their graph), the residual of K z = b is checked on every k-th solve only - the
  // reference's MA57 path never checks (fact_ma57.c:18).  1: every solve.
  int refine_check_every = 8;
  bool skip_resid_now = false;   // (the solve being queued is one of the unchecked ones)
  long solves_since_check = 0;
  bool decide_deferred = false;  // ... and such a verdict is outstanding
  bool ctl_pending = false;      // the control block of the last solve has not been looked at yet
  const double* last_b = nullptr;
  double* last_z = nullptr;
  RefineCtl last_ctl = {1, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0};
  int solve_seq = 0;             // solves with a residual queued since the control block was last cleared
  bool use_graph = true;         // replay captured hipGraphs instead of re-enqueueing ~100 launches
  // The single-launch dataflow kernels rely on workgroups being dispatched in index order (observed on every
  // gfx9 part, documented nowhere).  Their spins are bounded; should one ever time out, the handle falls back for
  // good to the per-level launches (same arithmetic, no cross-workgroup waits) and repeats the work at once.
  bool no_dataflow = false;
  int fake_timeouts = 0;         // test hook: the next reads of the info words report a timeout
  long dataflow_fallbacks = 0;
  int debug_phases = 15;         // timing-only phase mask of k_factor_level (15 = everything)
  int split_max_fronts = 1 << 30;  // levels with at most this many fronts use the split kernels
  int solve_whole_max = 48;   // a front stays ONE solve item while an item thread holds at most this many panel entries (SOLVE_PREFETCH of them before its wait)
  bool cg_graph = false;      // chunks of the device-controlled CG as captured graphs (slower than direct launches here)
  double* x_dot_out = nullptr;  // set around a projection of the device-controlled CG
  int x_dot_blocks = 0;
  int xupd_blocks = 256;      // workgroups of the x update inside the tree launch (xupd_fused)
  bool solve_sorted = true;   // solve items of a level: biggest fronts first
  int factor_top_max = 160;   // levels with at most this many fronts join the single-launch top-of-tree factorisation (0: off)
  int factor_top_levels = 1 << 20;  // at most this many levels in the single-launch top-of-tree factorisation (tests)
  int factor_top_fine = 12;   // levels with at most this many fronts use finer panel / Schur items there
  bool zero_behind = false;   // solve-panel items put the panels of the bottom levels back to zero (see l_prefix): -10 us on the fill, +15 us on the dataflow launch (measured), off
  int factor_top_post = 64;   // levels with at most this many fronts post the pivot block to polling panel workgroups
  int wide_min_rows = 256;    // fronts with at least this many update rows are solved by several workgroups (0: off; one workgroup streams a panel at ~50 GB/s)
  int top_prefetch = 1;       // top-of-tree solve kernels prefetch their panels before the dependency wait
  int panel_small_below = 0;  // levels with fewer 128-row panel blocks use 64-row blocks
  int pull_max_children = 4;  // 0: always the separate assembly kernel; otherwise pull for any number of children
  int top_max_fronts = 1024;
  long cache_hits = 0, analyses = 0, num_factor = 0, num_solve = 0;
  int info_host[INFO_WORDS] = {0, 0, 0, 0};
  Prof prof;
  // shared by all plan states (fixed size, never reallocated: graphs of every state may point at them)
  DevBuf d_info, d_norms, d_ctl;
  PinBuf h_ctl;
  void* h_ctl_dev = nullptr;  // device address of the pinned copy of the control block
  PinBuf h_stage, h_info;
  DevBuf d_sp_idx, d_sp_val;
  // assembly
  DevBuf d_jp, d_ji, d_jx, d_vi, d_ci, d_cnt, d_akp, d_aki, d_akx;
  // projected CG
  DevBuf d_cg_b, d_cg_z, d_cg_vec, d_cg_dots, d_cg_ctl;
  PinBuf h_cg_dots, h_hv, h_cg_ctl;
  bool cg_device_loop = true;  // projected CG with the loop control on the device (krylov_device.inc)
  long cg_device_runs = 0, cg_device_fallbacks = 0;
  DevBuf d_lz_Q, d_lz_b, d_lz_coef;  // generalised Lanczos: basis (n x cap), three rotating right-hand sides, coefficients
};

struct hipfact_spmat {
  hipfact_handle* h = nullptr;
  int rows = 0, cols = 0;
  long long nnz = 0;
  DevBuf cp, ri, val;      // CSC as given (= CSR of M^T)
  DevBuf tp, ti, tval, tsrc;  // CSR of M
  DevBuf dx, dy;
};

#define HCHECK(h, call)                                                                     \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      (h)->error = std::string(#call) + ": " + hipGetErrorString(e__);                      \
      return e__ == hipErrorOutOfMemory ? HIPFACT_ENOMEM : HIPFACT_EDEVICE;                 \
    }                                                                                       \
  } while (0)

static inline int nblocks(long long n, int cap = 4096) {
  long long b = (n + FB - 1) / FB;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

static void prof_begin(hipfact_handle* h, int cls) {
  Prof& p = h->prof;
  if (!p.on) return;
  if (p.used + 2 > p.ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    p.ev.push_back(a);
    p.ev.push_back(b);
  }
  p.cls.resize(p.ev.size() / 2);
  p.cls[p.used / 2] = cls;
  (void)hipEventRecord(p.ev[p.used], h->stream);
}
static void prof_end(hipfact_handle* h) {
  Prof& p = h->prof;
  if (!p.on || p.used + 2 > p.ev.size()) return;
  (void)hipEventRecord(p.ev[p.used + 1], h->stream);
  p.used += 2;
}
static void prof_collect(hipfact_handle* h) {
  Prof& p = h->prof;
  if (p.used == 0) return;
  (void)hipStreamSynchronize(h->stream);
  for (size_t i = 0; i < p.used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) == hipSuccess) {
      p.ms[p.cls[i / 2]] += ms;
      p.cnt[p.cls[i / 2]] += 1;
    }
  }
  p.used = 0;
}
// launch wrapper: LAUNCH(class, kernel, grid, block, lds, args...)
#define LAUNCH(cls, kernel, grid, block, lds, ...)                         \
  do {                                                                     \
    prof_begin(h, cls);                                                    \
    hipLaunchKernelGGL(kernel, grid, block, lds, h->stream, __VA_ARGS__);  \
    prof_end(h);                                                           \
  } while (0)

template <class T>
static int upload(hipfact_handle* h, DevBuf& buf, const std::vector<T>& v) {
  HCHECK(h, buf.ensure(std::max<size_t>(v.size() * sizeof(T), 16)));
  if (!v.empty()) HCHECK(h, hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return HIPFACT_OK;
}

// LDS of the solve-panel builder (k_build_solve_panels / role 3 of k_factor_top): X | 1 / d | tiles | offsets
static size_t solve_panel_lds(int wmax) {
  const size_t wp = (size_t)((wmax + 15) & ~15);
  return (wp * (wp + 1) + wp + 8 * 16 * 17 + wp + 1024) * sizeof(double);
}

static int dense_upload(hipfact_handle* h, size_t vec_bytes);

static int upload_plan(hipfact_handle* h) {
  const Plan& P = h->plan;
  const int ns = P.nsuper;
  std::vector<SnDesc> sn(ns);
  for (int s = 0; s < ns; ++s) {
    SnDesc& d = sn[s];
    d.Loff = P.sn_Loff[s];
    d.Uoff = P.sn_Uoff[s];
    d.uoff = P.sn_uoff[s];
    d.rowoff = P.sn_rowptr[s];
    d.reloff = P.rel_ptr[s];
    d.c0 = P.sn_c0[s];
    d.w = P.sn_c0[s + 1] - P.sn_c0[s];
    d.r = P.sn_r[s];
    d.parent = P.sn_parent[s];
    d.child_begin = P.child_ptr[s];
    d.child_end = P.child_ptr[s + 1];
    d.pad0 = P.sn_level[s];
    d.pad1 = 0;
  }
  // inverse relative indices for the pull-mode extend-add: for child s with parent p,
  // inv[pad1(s) + q] = row of s's update matrix that lands on front row q of p, or -1
  std::vector<int> inv;
  {
    long long total = 0;
    for (int s = 0; s < ns; ++s)
      if (P.sn_parent[s] >= 0) total += P.sn_r[P.sn_parent[s]];
    if (total >= (1LL << 31)) {
      h->error = "inverse index map exceeds 2^31 entries";
      return HIPFACT_EINTERNAL;
    }
    inv.assign((size_t)total, -1);
    long long off = 0;
    for (int s = 0; s < ns; ++s) {
      const int p = P.sn_parent[s];
      if (p < 0) continue;
      sn[s].pad1 = (int)off;
      const int us = P.sn_r[s] - (P.sn_c0[s + 1] - P.sn_c0[s]);
      for (int a = 0; a < us; ++a) inv[(size_t)off + P.rel[P.rel_ptr[s] + a]] = a;
      off += P.sn_r[p];
    }
  }
  // children of every front in blocks of MAXCH: the first block travels inside the work items, the
  // others (fronts with more than MAXCH children) sit in an overflow array, chained by index
  std::vector<PullDesc> pulls(ns), pullx;
  for (int s = 0; s < ns; ++s) {
    const int nch = P.child_ptr[s + 1] - P.child_ptr[s];
    PullDesc* cur = &pulls[s];
    memset(cur, 0, sizeof(*cur));
    cur->next = -1;
    int last_x = -1;  // index of the block being filled inside pullx (-1: pulls[s])
    for (int k = 0; k < nch; ++k) {
      if (k > 0 && k % MAXCH == 0) {
        PullDesc nb;
        memset(&nb, 0, sizeof(nb));
        nb.next = -1;
        pullx.push_back(nb);
        const int idx = (int)pullx.size() - 1;
        (last_x < 0 ? pulls[s] : pullx[(size_t)last_x]).next = idx;
        last_x = idx;
      }
      PullDesc& pd = last_x < 0 ? pulls[s] : pullx[(size_t)last_x];
      const int ch = P.child_idx[P.child_ptr[s] + k];
      const int q = k % MAXCH;
      pd.Uoff[q] = sn[ch].Uoff;
      pd.reloff[q] = sn[ch].reloff;
      pd.invoff[q] = sn[ch].pad1;
      pd.uc[q] = sn[ch].r - sn[ch].w;
      pd.n = q + 1;
    }
  }
  int rc;
  if ((rc = upload(h, h->d_inv, inv))) return rc;
  if ((rc = upload(h, h->d_pullx, pullx))) return rc;
  if ((rc = upload(h, h->d_sn, sn))) return rc;
  if ((rc = upload(h, h->d_level_sn, P.level_sn))) return rc;
  if ((rc = upload(h, h->d_rows, P.sn_rows))) return rc;
  if ((rc = upload(h, h->d_rel, P.rel))) return rc;
  if ((rc = upload(h, h->d_child, P.child_idx))) return rc;
  h->idx32 = P.saddle && P.L_size < (1LL << 32) && P.nprod < (1LL << 32);
  if (h->idx32) {
    std::vector<unsigned int> t32(P.Mtarget.begin(), P.Mtarget.end());
    if ((rc = upload(h, h->d_Mtarget, t32))) return rc;
  } else if ((rc = upload(h, h->d_Mtarget, P.Mtarget)))
    return rc;
  if ((rc = upload(h, h->d_perm, P.perm))) return rc;
  if ((rc = upload(h, h->d_Kp, P.Kp))) return rc;
  if ((rc = upload(h, h->d_Ki, P.Ki))) return rc;
  if (P.saddle) {
    if (h->idx32) {
      std::vector<unsigned int> p32(P.prod_ptr.begin(), P.prod_ptr.end());
      if ((rc = upload(h, h->d_prod_ptr, p32))) return rc;
    } else if ((rc = upload(h, h->d_prod_ptr, P.prod_ptr)))
      return rc;
    {
      // one packed word per product when every pair fits (see prod_pair)
      h->prod_packed = P.nnzK < (1LL << 24);
      const size_t np = P.prod_a.size();
      for (size_t p = 0; p < np && h->prod_packed; ++p)
        if (std::abs(P.prod_a[p] - P.prod_b[p]) > 255) h->prod_packed = false;
      if (h->prod_packed) {
        std::vector<int> pk(np);
        for (size_t p = 0; p < np; ++p)
          pk[p] = (int)(((unsigned int)std::min(P.prod_a[p], P.prod_b[p]) << 8) |
                        (unsigned int)std::abs(P.prod_a[p] - P.prod_b[p]));
        if ((rc = upload(h, h->d_prod_a, pk))) return rc;
        h->d_prod_b.release();
      } else {
        if ((rc = upload(h, h->d_prod_a, P.prod_a))) return rc;
        if ((rc = upload(h, h->d_prod_b, P.prod_b))) return rc;
      }
    }
    if ((rc = upload(h, h->d_Ar_ptr, P.Ar_ptr))) return rc;
    if ((rc = upload(h, h->d_Ar_col, P.Ar_col))) return rc;
    if ((rc = upload(h, h->d_Ar_src, P.Ar_src))) return rc;
    if ((rc = upload(h, h->d_Kc_y, P.Kc_y))) return rc;
    HCHECK(h, h->d_Ar_val.ensure(std::max<size_t>(P.Ar_src.size() * sizeof(double), 16)));
  } else {
    if ((rc = upload(h, h->d_src, P.src))) return rc;
    // CSR of the lower triangle (row access for the symmetric residual)
    const int N = P.N;
    std::vector<int> Tp(N + 1, 0), Ti(P.nnzK), Tsrc(P.nnzK);
    for (int j = 0; j < N; ++j)
      for (int e = P.Kp[j]; e < P.Kp[j + 1]; ++e) ++Tp[P.Ki[e] + 1];
    for (int i = 0; i < N; ++i) Tp[i + 1] += Tp[i];
    std::vector<int> fill(Tp.begin(), Tp.end() - 1);
    for (int j = 0; j < N; ++j)
      for (int e = P.Kp[j]; e < P.Kp[j + 1]; ++e) {
        const int q = fill[P.Ki[e]]++;
        Ti[q] = j;
        Tsrc[q] = e;
      }
    if ((rc = upload(h, h->d_Tp, Tp))) return rc;
    if ((rc = upload(h, h->d_Ti, Ti))) return rc;
    if ((rc = upload(h, h->d_Tsrc, Tsrc))) return rc;
  }
  // per-level launch metadata
  h->levels.assign(P.nlevels, LevelInfo());
  size_t max_lds = 0;
  std::vector<int> items;
  std::vector<FrontItem> fitems;
  for (int l = 0; l < P.nlevels; ++l) {
    LevelInfo& li = h->levels[l];
    li.begin = P.level_ptr[l];
    li.count = P.level_ptr[l + 1] - P.level_ptr[l];
    int mw = 0, mr = 0, mu = 0, mch = 0;
    double work = 0;
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
      const int s = P.level_sn[q];
      mch = std::max(mch, P.child_ptr[s + 1] - P.child_ptr[s]);
      const int w = P.sn_c0[s + 1] - P.sn_c0[s], r = P.sn_r[s];
      mw = std::max(mw, w);
      mr = std::max(mr, r);
      mu = std::max(mu, r - w);
      work = std::max(work, (double)r * r * w);
    }
    const size_t wp = (size_t)((mw + 15) & ~15);
    const size_t needB = wp * (wp + 1) + 16 * (wp + 1) + 32;
    const size_t needD = (size_t)128 * 32;  // two 64 x KC operand strips
    li.lds_factor = (wp + std::max(needB, needD)) * sizeof(double);
    li.lds_pivot = (wp + needB) * sizeof(double) + MAXCH * wp * sizeof(int);
    li.lds_panel = (wp + wp * (wp + 1)) * sizeof(double) + MAXCH * wp * sizeof(int);
    li.lds_schur = (wp + needD) * sizeof(double) + 128 * MAXCH * sizeof(int);
    {
      // scatter assembly stages one child's relative indices at a time: the largest child of this level
      int muc = 0;
      for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
        const int s = P.level_sn[q];
        for (int ci = P.child_ptr[s]; ci < P.child_ptr[s + 1]; ++ci) {
          const int ch = P.child_idx[ci];
          muc = std::max(muc, P.sn_r[ch] - (P.sn_c0[ch + 1] - P.sn_c0[ch]));
        }
      }
      li.lds_asm = ((size_t)muc + 16) * sizeof(int);
    }
    li.lds_fwd = ((size_t)mr + 9 * (size_t)mw + 1024 + 2) * sizeof(double);
    {
      // dev_bwd_front: u + w; dev_bwd_small (u <= 256): 4 ceil(u/4) + (wp + 4) + wp + 256 + ceil(u/2)
      const size_t us = (size_t)std::min(mu, 256);
      li.lds_bwd = std::max((size_t)mu + mw + 2, us + 4 + 2 * wp + 4 + 256 + us / 2 + 2) * sizeof(double);
    }
    li.lds_solve_max = std::max(li.lds_fwd, li.lds_bwd);
    max_lds = std::max({max_lds, li.lds_factor});
    // split when the level cannot fill the chip with one workgroup per front and the fronts are not tiny
    li.split = (li.count <= h->split_max_fronts) && (work >= 2.0e5);
    li.pull = li.split && mch > 0 && h->pull_max_children > 0;  // any number of children (descriptor chains)
    li.chain = mch > MAXCH;
    {
      // assembly items: (front, target-column class) for every front that has children
      int with_children = 0;
      for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q)
        with_children += (P.child_ptr[P.level_sn[q] + 1] > P.child_ptr[P.level_sn[q]]);
      // workgroups per front: enough to fill the chip for few fronts, and one per ~16 target
      // columns for very large fronts (dense Schur complements copy tens of MB per level)
      li.nparts = with_children > 0 ? std::max(1, std::min(32, 768 / with_children)) : 1;
      if (with_children > 0 && mr >= 1024) li.nparts = std::max(li.nparts, std::min(256, mr / 16));
      li.itA = (long long)items.size();
      for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
        const int s = P.level_sn[q];
        if (P.child_ptr[s + 1] == P.child_ptr[s]) continue;
        for (int p = 0; p < li.nparts; ++p) {
          items.push_back(s);
          items.push_back(p);
          ++li.nA;
        }
      }
    }
    if (li.split) {
      auto item = [&](int s, int part) {
        FrontItem it;
        memset(&it, 0, sizeof(it));
        it.Loff = sn[s].Loff;
        it.Uoff = sn[s].Uoff;
        it.w = sn[s].w;
        it.r = sn[s].r;
        it.part = part;
        it.nchild = sn[s].child_end - sn[s].child_begin;
        it.pd = pulls[s];
        return it;
      };
      // widest fronts first: a level lasts as long as its slowest front, which must not be the
      // one that had to wait for a free CU
      std::vector<int> order(P.level_sn.begin() + P.level_ptr[l], P.level_sn.begin() + P.level_ptr[l + 1]);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const long long wa = sn[a].w, wb = sn[b].w;
        return wa != wb ? wa > wb : sn[a].r > sn[b].r;
      });
      li.itB = (long long)fitems.size();
      for (int s : order) fitems.push_back(item(s, 0));
      li.itC = (long long)fitems.size();
      {
        // 128 panel rows per workgroup (8 waves); 64 when that leaves most of the chip idle
        long long blocks128 = 0;
        for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
          const int s = P.level_sn[q];
          blocks128 += (P.sn_r[s] - (P.sn_c0[s + 1] - P.sn_c0[s]) + 127) / 128;
        }
        li.panel_threads = (blocks128 < h->panel_small_below) ? 256 : 512;
      }
      const int prow = li.panel_threads / 4;
      for (int s : order) {
        const int u = P.sn_r[s] - (P.sn_c0[s + 1] - P.sn_c0[s]);
        for (int b = 0; b < (u + prow - 1) / prow; ++b) {
          fitems.push_back(item(s, b));
          ++li.nC;
        }
      }
      li.itD = (long long)fitems.size();
      for (int s : order) {
        const int u = P.sn_r[s] - (P.sn_c0[s + 1] - P.sn_c0[s]);
        const int nt = (u + 63) / 64;
        for (int I = 0; I < nt; ++I)
          for (int J = 0; J <= I; ++J) {
            fitems.push_back(item(s, (I << 16) | J));
            ++li.nD;
          }
      }
    }
  }
  h->ent_fused = h->ent_split = h->rows_fused = h->rows_split = 0;
  for (int l = 0; l < P.nlevels; ++l)
    for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
      const int s = P.level_sn[q];
      const double w = P.sn_c0[s + 1] - P.sn_c0[s], r = P.sn_r[s];
      const double ent = w * r - w * (w - 1) / 2;
      (h->levels[l].split ? h->ent_split : h->ent_fused) += ent;
      (h->levels[l].split ? h->rows_split : h->rows_fused) += r;
    }
  if ((rc = upload(h, h->d_items, items))) return rc;
  if ((rc = upload(h, h->d_fitems, fitems))) return rc;
  // Items of the fused solve launch (level order).  An item keeps its share of the solve panel in registers across
  // its dependency wait: SOLVE_PREFETCH entries per thread, i.e. rows x w <= 1024 x SOLVE_PREFETCH per item (what
  // does not fit is loaded behind the wait, on the critical path of the tree: a memory round trip per entry).
  // A front with more rows than that is cut into row slices (slice 0: the pivot rows and the first update rows;
  // update rows in multiples of 16), see SolveItem.
  auto slice_rows = [&](int w2) {  // most rows an item of a front of width w2 can hold
    const int fwd = 1024 / ((w2 + SOLVE_PREFETCH - 1) / SOLVE_PREFETCH);  // Ef = ceil(w / floor(1024 / rows)) <= PREFETCH
    const int bwd = SOLVE_PREFETCH * std::max(1, 1024 / w2);              // Eb = ceil(rows / floor(1024 / w)) <= PREFETCH
    return std::max(w2 + 16, std::min({fwd, bwd, 1024}));
  };
  std::vector<int> it_front, it_sl, it_nsl, it_a0, it_a1, first_item(ns, 0);
  // Inside a level the biggest fronts come first (both sweeps): a level with more items than CUs runs in rounds (one
  // item per CU: its panel sits in registers), and the last round should be the cheap one.
  std::vector<int> solve_order(P.level_sn.begin(), P.level_sn.begin() + ns);
  if (h->solve_sorted)
    for (int l = 0; l < P.nlevels; ++l)
      std::stable_sort(solve_order.begin() + P.level_ptr[l], solve_order.begin() + P.level_ptr[l + 1], [&](int a, int b) {
        return (long long)sn[a].r * sn[a].w > (long long)sn[b].r * sn[b].w;
      });
  for (int q = 0; q < ns; ++q) {
    const int s2 = solve_order[q];
    const int w2 = std::max(1, sn[s2].w), u2 = sn[s2].r - sn[s2].w;
    const int cap = slice_rows(w2);
    first_item[s2] = (int)it_front.size();
    std::vector<std::pair<int, int>> cuts;
    // sliced when it must be (more than 1024 rows) or when more than half of an item's entries would be loaded behind
    // its wait (config 3's fronts of 600-1000 rows x 126 columns: 28 us per level); fronts a little over the register
    // capacity stay whole - a second item and the exchange of partial sums cost more than 18 loads (config 4)
    const int q1 = std::max(1, 1024 / std::max(1, sn[s2].r));
    const bool whole = sn[s2].r <= 1024 && (w2 + q1 - 1) / q1 <= h->solve_whole_max;
    if (sn[s2].r <= cap || whole || !h->solve_slices) {
      cuts.push_back({0, u2});
    } else {
      // (an item stages the pivot rows and its update rows, one per thread: w + rows <= 1024)
      const int first = std::max(16, (cap - w2) & ~15), rest = std::max(16, std::min(cap, 1024 - w2) & ~15);
      cuts.push_back({0, std::min(u2, first)});
      for (int a = cuts.back().second; a < u2; a += rest) cuts.push_back({a, std::min(u2, a + rest)});
    }
    for (size_t sl = 0; sl < cuts.size(); ++sl) {
      it_front.push_back(s2);
      it_sl.push_back((int)sl);
      it_nsl.push_back((int)cuts.size());
      it_a0.push_back(cuts[sl].first);
      it_a1.push_back(cuts[sl].second);
    }
  }
  const int nit = (int)it_front.size();
  {
    // top-of-tree factorisation in one launch: the last levels, as long as every one of them is
    // narrow and can pull its extend-add (no front with more than MAXCH children)
    h->ftop_level = 1 << 30;
    h->ftop_count = 0;
    int lvl = P.nlevels;
    while (lvl > 0) {
      const LevelInfo& li = h->levels[lvl - 1];
      int mch = 0;
      for (int q = P.level_ptr[lvl - 1]; q < P.level_ptr[lvl]; ++q)
        mch = std::max(mch, P.child_ptr[P.level_sn[q] + 1] - P.child_ptr[P.level_sn[q]]);
      if (!(li.count <= h->factor_top_max && h->pull_max_children > 0 && mch <= MAXCH && P.nlevels - lvl < h->factor_top_levels)) break;
      --lvl;
    }
    h->sp_folded = false;
    h->mini_x_bytes = 0;
    if (P.nlevels - lvl >= 2 && h->factor_top_max > 0) {
      // fused solve: will there be solve panels (same conditions as below), and are they built inside this launch?
      size_t sp_lds_pre = 0;
      bool fold = h->solve_fused && h->spanel_fold && ns > 0;
      {
        int wmax = 1;
        for (int s2 = 0; s2 < ns; ++s2) {
          fold = fold && sn[s2].w >= 1 && (h->solve_slices || sn[s2].r <= 1024);
          wmax = std::max(wmax, sn[s2].w);
        }
        sp_lds_pre = solve_panel_lds(wmax);
        fold = fold && sp_lds_pre <= 160 * 1024;
      }
      auto ntiles = [&](int s) {
        const long long u = sn[s].r - sn[s].w, nt = (u + 63) / 64;
        return nt * (nt + 1) / 2;
      };
      auto ntiles32 = [&](int s) {
        const long long u = sn[s].r - sn[s].w, nt = (u + 31) / 32;
        return nt * (nt + 1) / 2;
      };
      // panel rows per workgroup.  Levels that post the pivot block solve their panels by block substitution, one
      // wave per strip of 16 rows and ONE computing wave per SIMD (64 rows per workgroup, the other four waves only
      // poll and stage): the substitution is bound by the fp64 matrix pipe, and with two computing waves per SIMD a
      // workgroup falls behind the pivot workgroup it follows
      auto panel_rows = [&](int count) {
#ifdef HIPFACT_PIVOT_V1
        return count <= h->factor_top_fine ? 64 : 128;
#else
        return (count <= h->factor_top_post || count <= h->factor_top_fine) ? 64 : 128;
#endif
      };
      std::vector<TopFItem> tf;
      std::vector<size_t> level_end, panel_end;  // end of the items / of the pivot and panel items of each level in tf
      size_t lds = 0;
      // slots of the posted pivot blocks (wp x wp each), all sentinel between factorisations
      std::vector<long long> xoff(ns, 0);
      long long xsize = 0;
      for (int l = lvl; l < P.nlevels; ++l)
        for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
          const int s = P.level_sn[q];
          const long long wp = (sn[s].w + 15) & ~15;
          xoff[s] = xsize;
          xsize += wp * wp;
        }
      // ... and of the single-front levels below the launch that run their pivot and panel items as a small
      // dataflow launch of their own (dense chains): these slots are put back by a fill per factorisation
      h->mini_x_off = xsize;
      for (int l = 0; l < lvl; ++l) {
        LevelInfo& lm = h->levels[l];
        lm.mini_cnt = 0;
        if (!(h->chain_fuse && lm.split && lm.pull && !lm.chain && lm.count == 1)) continue;
        const int s = P.level_sn[P.level_ptr[l]];
        if (sn[s].r - sn[s].w < 512 || sn[s].child_end - sn[s].child_begin > MAXCH) continue;
        const long long wp = (sn[s].w + 15) & ~15;
        xoff[s] = xsize;
        xsize += wp * wp;
        lm.mini_cnt = -1;  // marked; items below
      }
      h->mini_x_bytes = (size_t)(xsize - h->mini_x_off) * sizeof(double);
      HCHECK(h, h->d_xarena.ensure(std::max<size_t>((size_t)xsize * sizeof(double), 16)));
      HCHECK(h, hipMemsetAsync(h->d_xarena.p, 0xFF, std::max<size_t>((size_t)xsize * sizeof(double), 16), h->stream));
      for (int l = lvl; l < P.nlevels; ++l) {
        const LevelInfo& li = h->levels[l];
        lds = std::max({lds, li.lds_pivot, li.lds_panel});
        // few fronts: workgroups are plentiful, so finer items shorten the per-level chain
        const bool fine = li.count <= h->factor_top_fine;
        auto base = [&](int s, int role, int part) {
          TopFItem t;
          memset(&t, 0, sizeof(t));
          t.it.Loff = sn[s].Loff;
          t.it.Uoff = sn[s].Uoff;
          t.it.w = sn[s].w;
          t.it.r = sn[s].r;
          t.it.part = part;
          t.it.nchild = sn[s].child_end - sn[s].child_begin;
          t.it.pd = pulls[s];
          t.role = role;
          t.front = s;
          t.part2 = part;
          t.nwait = t.it.nchild;
          for (int k = 0; k < t.nwait; ++k) {
            const int ch = P.child_idx[sn[s].child_begin + k];
            t.wait_id[k] = ch;
            t.wait_cnt[k] = P.sn_level[ch] >= lvl ? (int)(h->levels[P.sn_level[ch]].count <= h->factor_top_fine ? ntiles32(ch) : (ntiles(ch) + 1) / 2) : 0;
          }
          t.crows = fine ? 64 : 128;
          t.post = li.count <= h->factor_top_post;
          t.prows = panel_rows(li.count);
          t.xoff = xoff[s];
          t.target = (sn[s].r - sn[s].w + t.prows - 1) / t.prows;
          return t;
        };
        // Workgroups are dispatched in index order and a waiting one keeps its CU.  Pivot items:
        // widest front first (the level lasts as long as its slowest pivot).  Panel and Schur
        // items: narrowest front first - their pivots finish first, so on a level with more
        // workgroups than CUs the early slots go to work that is about to become ready.
        std::vector<int> wide_first(P.level_sn.begin() + P.level_ptr[l], P.level_sn.begin() + P.level_ptr[l + 1]);
        std::stable_sort(wide_first.begin(), wide_first.end(), [&](int a, int b) {
          return sn[a].w != sn[b].w ? sn[a].w > sn[b].w : sn[a].r > sn[b].r;
        });
        std::vector<int> narrow_first(wide_first.rbegin(), wide_first.rend());
        for (int s : wide_first) {
          tf.push_back(base(s, 0, 0));
          const size_t wp = (size_t)((sn[s].w + 15) & ~15);
          lds = std::max(lds, (wp + 2 * (size_t)(2 * 64 * 32 + 64 * MAXCH)) * sizeof(double));
        }
        for (int s : narrow_first) {
          const int u = sn[s].r - sn[s].w;
          const int crows = panel_rows(li.count);
          for (int b = 0; b < (u + crows - 1) / crows; ++b) tf.push_back(base(s, 1, b));
        }
        panel_end.push_back(tf.size());
        for (int s : narrow_first) {
          const int u = sn[s].r - sn[s].w, nt = fine ? (u + 31) / 32 : (u + 63) / 64;  // fine: 32 x 32 tiles
          std::vector<int> tiles;
          for (int I = 0; I < nt; ++I)
            for (int J = 0; J <= I; ++J) tiles.push_back((I << 16) | J);
          const int stride = fine ? 1 : 2;
          for (size_t k = 0; k < tiles.size(); k += stride) {
            TopFItem t = base(s, 2, tiles[k]);
            t.part2 = fine ? -1 : tiles[std::min(k + 1, tiles.size() - 1)];
            t.sidx = (int)(k / stride);
            t.scount = (int)((tiles.size() + stride - 1) / stride);
            tf.push_back(t);
          }
        }
        level_end.push_back(tf.size());
      }
      if (fold) {
        // Solve-panel items (role 3) dealt into the levels.  Workgroups are dispatched in index order and a
        // waiting one keeps its CU, so the residents are always the lowest-indexed unfinished items.  The Schur
        // items of a level cannot start before its pivot and panel items are through (~30 us): solve-panel items
        // (~20 us each) go between the panel and the Schur items of the level, as many as fit beside its pivot and
        // panel items - they run on CUs that the Schur items would only have occupied waiting.
        // Fronts in level order: those below the launch are final already, a front of the launch can follow the
        // panel items of a later level (or wait a moment for its own).  What is left goes behind the root.
        std::vector<TopFItem> out;
        out.reserve(tf.size() + nit);
        int qnext = 0;
        auto filler = [&](int q) {
          const int s = it_front[q];
          TopFItem t;
          memset(&t, 0, sizeof(t));
          t.it.Loff = sn[s].Loff;
          t.it.w = sn[s].w;
          t.it.r = sn[s].r;
          t.it.part = q;  // index of the SolveItem (level order; a sliced front has several)
          t.role = 3;
          t.front = s;
          t.part2 = -1;
          if (P.sn_level[s] >= lvl) {
            const int crows = panel_rows(h->levels[P.sn_level[s]].count);
            t.nwait = 1;
            t.target = (sn[s].r - sn[s].w + crows - 1) / crows;
          }
          return t;
        };
        // a front of the launch without update rows (the root) writes its solve panel in its pivot item
        std::vector<char> own(ns, 0);
        for (TopFItem& t : tf)
          if (t.role == 0 && t.it.r == t.it.w) {
            t.sidx = first_item[t.front] + 1;
            own[t.front] = 1;
          }
        auto skip_own = [&] {
          while (qnext < nit && own[it_front[qnext]]) ++qnext;
        };
        size_t from = 0;
        for (int l = lvl; l < P.nlevels; ++l) {
          const size_t mid = panel_end[l - lvl], to = level_end[l - lvl];
          out.insert(out.end(), tf.begin() + from, tf.begin() + mid);
          long long room = (long long)h->spanel_fold_room - (long long)(mid - from);
          for (skip_own(); room > 0 && qnext < nit && P.sn_level[it_front[qnext]] <= l; skip_own()) {
            out.push_back(filler(qnext++));
            --room;
          }
          out.insert(out.end(), tf.begin() + mid, tf.begin() + to);
          from = to;
        }
        for (skip_own(); qnext < nit; skip_own()) out.push_back(filler(qnext++));
        tf.swap(out);
        lds = std::max(lds, sp_lds_pre);
        h->sp_folded = true;
      }
      h->ftop_level = lvl;
      h->ftop_count = (int)tf.size();
      h->ftop_lds = lds;
      h->l_prefix = 0;
      if (h->sp_folded && lvl > 0 && lvl < P.nlevels && h->zero_behind) {
        bool whole = true;  // (a sliced front has several solve-panel items that read its pivot block: not handled)
        for (int q2 = 0; q2 < nit && whole; ++q2) whole = P.sn_level[it_front[q2]] >= lvl || it_nsl[q2] == 1;
        if (whole) h->l_prefix = P.sn_Loff[P.level_sn[P.level_ptr[lvl]]];
      }
      for (int l = 0; l < lvl; ++l) {
        LevelInfo& lm = h->levels[l];
        if (lm.mini_cnt == 0) continue;
        const int s = P.level_sn[P.level_ptr[l]];
        auto mini = [&](int role, int part) {
          TopFItem t;
          memset(&t, 0, sizeof(t));
          t.it.Loff = sn[s].Loff;
          t.it.Uoff = sn[s].Uoff;
          t.it.w = sn[s].w;
          t.it.r = sn[s].r;
          t.it.part = part;
          t.it.nchild = sn[s].child_end - sn[s].child_begin;
          t.it.pd = pulls[s];
          t.role = role;
          t.front = s;
          t.part2 = 0;
          t.nwait = t.it.nchild;  // the children finished in earlier launches: nothing to wait for
          for (int k = 0; k < t.nwait; ++k) t.wait_id[k] = P.child_idx[sn[s].child_begin + k];
          t.crows = 64;
          t.prows = panel_rows(1);
          t.post = 1;
          t.xoff = xoff[s];
          t.target = (sn[s].r - sn[s].w + t.prows - 1) / t.prows;
          return t;
        };
        lm.mini_off = (long long)tf.size();
        tf.push_back(mini(0, 0));
        for (int b = 0; b < (sn[s].r - sn[s].w + panel_rows(1) - 1) / panel_rows(1); ++b) tf.push_back(mini(1, b));
        lm.mini_cnt = (int)(tf.size() - (size_t)lm.mini_off);
        lm.mini_lds = std::max(lm.lds_pivot, lm.lds_panel);
      }
      if ((rc = upload(h, h->d_tfitems, tf))) return rc;
    }
  }
  {
    // levels merged into the single-launch top-of-tree solve: as many of the last levels as fit
    // the co-residency cap, and only if that saves at least two launches
    int lvl = P.nlevels, total = 0;
    while (lvl > 0 && total + h->levels[lvl - 1].count <= h->top_max_fronts) total += h->levels[--lvl].count;
    h->top_level = 1 << 30;
    h->top_count = 0;
    if (P.nlevels - lvl >= 3 && h->top_max_fronts > 0) {
      h->top_level = lvl;
      h->top_count = total;
      std::vector<int> top;
      std::vector<TopItem> titems;
      h->top_lds_fwd = h->top_lds_bwd = 0;
      // wide fronts: a head and slices of WIDE_SLICE_ROWS update rows; the flag of such a front
      // counts its slices in the forward pass
      std::vector<int> ftarget(ns, 1);
      auto wide_slices = [&](int s) {
        const long long u = sn[s].r - sn[s].w;
        const int nch = sn[s].child_end - sn[s].child_begin;
        return (h->wide_min_rows > 0 && u >= h->wide_min_rows && nch <= MAXCH)
                   ? (int)((u + WIDE_SLICE_ROWS - 1) / WIDE_SLICE_ROWS)
                   : 0;
      };
      for (int l = lvl; l < P.nlevels; ++l)
        for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
          const int nsl = wide_slices(P.level_sn[q]);
          if (nsl > 0) ftarget[P.level_sn[q]] = nsl;
        }
      long long wpart_size = 0;
      for (int l = lvl; l < P.nlevels; ++l) {
        for (int q = P.level_ptr[l]; q < P.level_ptr[l + 1]; ++q) {
          const int s = P.level_sn[q];
          top.push_back(s);
          TopItem T;
          memset(&T, 0, sizeof(T));
          T.Loff = sn[s].Loff;
          T.uoff = sn[s].uoff;
          T.rowoff = sn[s].rowoff;
          T.s = s;
          T.c0 = sn[s].c0;
          T.w = sn[s].w;
          T.r = sn[s].r;
          T.parent = sn[s].parent;
          const int nch = sn[s].child_end - sn[s].child_begin;
          const long long u = T.r - T.w;
          T.nchild = nch <= MAXCH ? nch : -1;
          for (int k = 0; k < nch && nch <= MAXCH; ++k) {
            const int ch = P.child_idx[sn[s].child_begin + k];
            T.c_uoff[k] = sn[ch].uoff;
            T.c_reloff[k] = sn[ch].reloff;
            T.c_uc[k] = sn[ch].r - sn[ch].w;
            T.c_id[k] = ch;
            T.c_invoff[k] = sn[ch].pad1;
            T.c_wait[k] = P.sn_level[ch] >= lvl ? ftarget[ch] : 0;
          }
          size_t lf = h->levels[l].lds_fwd, lb = h->levels[l].lds_bwd;
          if (h->top_prefetch && T.nchild >= 0 && T.r <= 1024 && u * T.w <= TOP_L21_CAP) {  // one front row per thread
            T.prefetch |= 1;
            lf = ((size_t)T.r + 9 * (size_t)T.w + 1024 + TOP_REL_CAP / 2 + (size_t)(u * T.w) + 2) * sizeof(double);
          }
          if (u <= 256) {  // same arithmetic as the level kernels' small-front path
            T.prefetch |= 2;
            const size_t wp16 = (size_t)((T.w + 15) & ~15);
            lb = ((size_t)u + 4 + 2 * wp16 + 4 + 256 + (size_t)((u + 1) / 2) + (size_t)T.w * T.w + 2) * sizeof(double);
          }
          const int nsl = wide_slices(s);
          if (nsl > 0) {
            T.prefetch = 0;
            T.kind = 1;
            T.nsl = nsl;
            T.poff = wpart_size;
            wpart_size += (long long)nsl * T.w;
            lf = lb = ((size_t)10 * T.w + WIDE_SLICE_ROWS + 1024 + 2 + (size_t)nsl * T.w) * sizeof(double);
            titems.push_back(T);
            for (int q2 = 0; q2 < nsl; ++q2) {
              TopItem S2 = T;
              S2.kind = 2;
              S2.a0 = q2 * WIDE_SLICE_ROWS;
              S2.a1 = (int)std::min<long long>(u, (long long)(q2 + 1) * WIDE_SLICE_ROWS);
              titems.push_back(S2);
            }
          } else {
            titems.push_back(T);
          }
          h->top_lds_fwd = std::max(h->top_lds_fwd, lf);
          h->top_lds_bwd = std::max(h->top_lds_bwd, lb);
        }
      }
      h->top_count = (int)titems.size();
      if ((rc = upload(h, h->d_ftarget, ftarget))) return rc;
      HCHECK(h, h->d_wpart.ensure(std::max<size_t>((size_t)wpart_size * sizeof(double), 16)));
      // (polled by the heads of the sliced fronts: sentinel between solves)
      HCHECK(h, hipMemsetAsync(h->d_wpart.p, 0xFF, std::max<size_t>((size_t)wpart_size * sizeof(double), 16), h->stream));
      if ((rc = upload(h, h->d_top_sn, top))) return rc;
      if ((rc = upload(h, h->d_titems, titems))) return rc;
    }
    HCHECK(h, h->d_flags.ensure(std::max<size_t>((size_t)4 * ns * sizeof(int), 16)));
    HCHECK(h, hipMemsetAsync(h->d_flags.p, 0, (size_t)4 * ns * sizeof(int), h->stream));
  }
  {
    // fused solve launch: every front (or row slice of a big front) as a SolveItem, children before parents
    h->fused_solve = false;
    h->n_sitems = 0;
    bool ok = h->solve_fused && ns > 0;
    for (int s2 = 0; s2 < ns && ok; ++s2) ok = sn[s2].w >= 1 && (h->solve_slices || sn[s2].r <= 1024);
    if (ok) {
      std::vector<SolveItem> si;
      std::vector<long long> xuoff;  // children beyond the first MAXCH of a front
      std::vector<int> xinvoff;
      std::vector<int> fx0(ns, 0), fx1(ns, 0);
      std::vector<long long> fpoff(ns, 0);
      si.reserve(nit);
      long long spf = 0, spb = 0, spart = 0;
      int wmax = 1;
      for (int q = 0; q < nit; ++q) {
        const int s2 = it_front[q], sl = it_sl[q], nsl = it_nsl[q];
        SolveItem T;
        memset(&T, 0, sizeof(T));
        T.c0 = sn[s2].c0;
        T.w = sn[s2].w;
        T.r = sn[s2].r;
        T.uoff = sn[s2].uoff;
        T.rowoff = sn[s2].rowoff;
        T.Loff = sn[s2].Loff;
        const int nch = sn[s2].child_end - sn[s2].child_begin;
        T.nchild = std::min(nch, MAXCH);
        if (sl == 0) {
          fx0[s2] = (int)xuoff.size();
          fpoff[s2] = spart;
          spart += (long long)(nsl - 1) * T.w;
        }
        for (int k = 0; k < nch; ++k) {
          const int ch = P.child_idx[sn[s2].child_begin + k];
          if (k < MAXCH) {
            T.c_uoff[k] = sn[ch].uoff;
            T.c_invoff[k] = sn[ch].pad1;
          } else if (sl == 0) {
            xuoff.push_back(sn[ch].uoff);
            xinvoff.push_back(sn[ch].pad1);
          }
        }
        if (sl == 0) fx1[s2] = (int)xuoff.size();
        T.xbegin = fx0[s2];
        T.xend = fx1[s2];
        T.sl = sl;
        T.nsl = nsl;
        T.poff = fpoff[s2];
        T.a0 = it_a0[q];
        T.a1 = it_a1[q];
        const int ro = (sl == 0 ? T.w : 0) + (T.a1 - T.a0);  // rows of the item's copies of S
        T.Qf = std::max(1, std::min(T.w, 1024 / ro));
        T.Ef = (T.w + T.Qf - 1) / T.Qf;
        T.Pb = std::max(1, std::min(ro, 1024 / T.w));
        T.Eb = (ro + T.Pb - 1) / T.Pb;
        T.spf = spf;
        T.spb = spb;
        spf += ((long long)T.Ef * ro * T.Qf + 1) & ~1LL;
        spb += ((long long)T.Eb * T.w * T.Pb + 1) & ~1LL;
        wmax = std::max(wmax, T.w);
        si.push_back(T);
      }
      h->sp_lds = solve_panel_lds(wmax);
      // items [nit, 2 nit): the backward sweep's order - levels from the root down, inside a level the same order as
      // forward (biggest first), the slices of a front last to first (slice 0 adds the others' partial sums)
      {
        std::vector<int> lvl_begin(P.nlevels + 1, nit);
        for (int q = nit - 1; q >= 0; --q) lvl_begin[P.sn_level[it_front[q]]] = q;
        for (int l = P.nlevels - 1; l >= 0; --l)
          if (lvl_begin[l] > lvl_begin[l + 1]) lvl_begin[l] = lvl_begin[l + 1];
        for (int l = P.nlevels - 1; l >= 0; --l)
          for (int q = lvl_begin[l]; q < lvl_begin[l + 1];) {
            const int nsl = it_nsl[q];
            for (int k = nsl - 1; k >= 0; --k) {
              const SolveItem t = si[q + k];
              si.push_back(t);
            }
            q += nsl;
          }
      }
      if (h->sp_lds <= 160 * 1024) {
        if ((rc = upload(h, h->d_sitems, si))) return rc;
        if ((rc = upload(h, h->d_sxuoff, xuoff))) return rc;
        if ((rc = upload(h, h->d_sxinvoff, xinvoff))) return rc;
        HCHECK(h, h->d_SPf.ensure(std::max<size_t>((size_t)spf * sizeof(double), 16)));
        HCHECK(h, h->d_SPb.ensure(std::max<size_t>((size_t)spb * sizeof(double), 16)));
        HCHECK(h, hipMemsetAsync(h->d_SPf.p, 0, std::max<size_t>((size_t)spf * sizeof(double), 16), h->stream));
        HCHECK(h, hipMemsetAsync(h->d_SPb.p, 0, std::max<size_t>((size_t)spb * sizeof(double), 16), h->stream));
        HCHECK(h, h->d_xhat.ensure(std::max<size_t>((size_t)P.m * sizeof(double), 16)));
        HCHECK(h, hipMemsetAsync(h->d_xhat.p, 0xFF, std::max<size_t>((size_t)P.m * sizeof(double), 16), h->stream));
        // partial sums of the backward items of sliced fronts: polled, sentinel between solves
        HCHECK(h, h->d_spart.ensure(std::max<size_t>((size_t)spart * sizeof(double), 16)));
        HCHECK(h, hipMemsetAsync(h->d_spart.p, 0xFF, h->d_spart.bytes, h->stream));
        HCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_solve_panels),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->fused_solve = true;
        h->n_sitems = nit;
        h->sp_bytes = (double)(spf + spb) * sizeof(double);
      }
    }
    if (h->sp_folded && !h->fused_solve) {
      h->error = "solve-panel items without solve panels";
      return HIPFACT_EINTERNAL;
    }
  }
  // Capacity limits of the LDS-resident working sets (documented in INTEGRATION.md).  Only what can actually
  // run is checked: the per-level solve kernels below the single-launch top (whose wide fronts are sliced and
  // need no front-sized buffer), the top kernels' own requirements, and the scatter assembly.
  {
    size_t solve_lds = 0, asm_lds = 0;
    if (!h->fused_solve) {
      const int ltop = std::min(h->top_level, P.nlevels);
      for (int l = 0; l < ltop; ++l) solve_lds = std::max(solve_lds, h->levels[l].lds_solve_max);
      if (ltop < P.nlevels) solve_lds = std::max({solve_lds, h->top_lds_fwd, h->top_lds_bwd});
    }
    for (int l = 0; l < P.nlevels; ++l) asm_lds = std::max(asm_lds, h->levels[l].lds_asm);
    char buf[200];
    if (solve_lds > 160 * 1024) {
      snprintf(buf, sizeof buf, "front too large for the LDS-resident solve vectors (%zu KB needed, 160 KB available; "
               "fronts of up to 1024 rows use the fused solve launch instead)", solve_lds >> 10);
      h->error = buf;
      return HIPFACT_EINTERNAL;
    }
    if (max_lds > 160 * 1024 || h->ftop_lds > 160 * 1024 || asm_lds > 160 * 1024) {
      snprintf(buf, sizeof buf, "front too large for the LDS-resident factorisation buffers (%zu KB needed)",
               std::max({max_lds, h->ftop_lds, asm_lds}) >> 10);
      h->error = buf;
      return HIPFACT_EINTERNAL;
    }
  }
  HCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_assemble),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (const void* fn : {reinterpret_cast<const void*>(k_front_pivot<false>), reinterpret_cast<const void*>(k_front_pivot<true>),
                         reinterpret_cast<const void*>(k_front_panel<false>), reinterpret_cast<const void*>(k_front_panel<true>),
                         reinterpret_cast<const void*>(k_front_schur<false>), reinterpret_cast<const void*>(k_front_schur<true>), reinterpret_cast<const void*>(k_factor_top),
                         reinterpret_cast<const void*>(k_fwd_top),
                         reinterpret_cast<const void*>(k_bwd_top)})
    HCHECK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_factor_level),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_level),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_level),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // numeric workspaces
  HCHECK(h, h->d_Kval.ensure(std::max<size_t>((size_t)P.nnzK * sizeof(double), 16)));
  HCHECK(h, h->d_L.ensure((size_t)P.L_size * sizeof(double) + (size_t)3 * P.nsuper * sizeof(int) + 32));
  HCHECK(h, h->d_U.ensure(std::max<size_t>((size_t)P.U_size * sizeof(double), 16)));
  HCHECK(h, h->d_uvec.ensure(std::max<size_t>((size_t)P.u_size * sizeof(double), 16)));
  HCHECK(h, h->d_y.ensure(std::max<size_t>((size_t)P.m * sizeof(double), 16)));
  // exchanged element by element in the single-launch solve sweeps: every slot starts as the
  // sentinel (all bits set) and is put back by the opposite sweep after use
  HCHECK(h, h->d_ysol.ensure(std::max<size_t>((size_t)2 * P.m * sizeof(double), 16)));  // two copies: the fused launches alternate
  HCHECK(h, hipMemsetAsync(h->d_uvec.p, 0xFF, std::max<size_t>((size_t)P.u_size * sizeof(double), 16), h->stream));
  HCHECK(h, hipMemsetAsync(h->d_ysol.p, 0xFF, std::max<size_t>((size_t)2 * P.m * sizeof(double), 16), h->stream));
  HCHECK(h, h->d_epoch.ensure(16));
  HCHECK(h, hipMemsetAsync(h->d_epoch.p, 0, 16, h->stream));
  const size_t nb = std::max<size_t>((size_t)std::max(P.N, h->N_ext) * sizeof(double), 16);
  HCHECK(h, h->d_rhs.ensure(nb));
  HCHECK(h, h->d_sol.ensure(nb));
  HCHECK(h, h->d_res.ensure(nb));
  {
    const int drc = dense_upload(h, nb);
    if (drc) return drc;
  }
  HCHECK(h, h->d_info.ensure(INFO_BYTES));  // info words + pivot min / max
  HCHECK(h, h->d_norms.ensure(3 * sizeof(double) * 4096));
  HCHECK(h, h->h_info.ensure(INFO_WORDS * sizeof(int) + 2 * 64 * sizeof(double)));
  if (!h->d_ctl.p) {
    HCHECK(h, h->d_ctl.ensure(sizeof(RefineCtl)));
    HCHECK(h, hipMemsetAsync(h->d_ctl.p, 0, sizeof(RefineCtl), h->stream));
  }
  if (!h->h_ctl.p) {
    HCHECK(h, h->h_ctl.ensure(sizeof(RefineCtl)));
    memset(h->h_ctl.p, 0, sizeof(RefineCtl));
    HCHECK(h, hipHostGetDevicePointer(&h->h_ctl_dev, h->h_ctl.p, 0));
  }
  if (P.saddle) {
    HCHECK(h, h->d_Ksc.ensure(std::max<size_t>((size_t)P.nnzK * sizeof(double), 16)));
    HCHECK(h, h->d_dscale.ensure(std::max<size_t>((size_t)P.m * sizeof(double), 16)));
  }
  return HIPFACT_OK;
}

static inline unsigned long long* minmax_ptr(const hipfact_handle* h) {
  return reinterpret_cast<unsigned long long*>(h->d_info.as<char>() + INFO_WORDS * sizeof(int));
}

static inline SaddleMaps saddle_maps(const hipfact_handle* h);
#include "dense_cols.inc"

// this factorisation's solve-panel items zero the panels of the bottom levels behind them
static bool zero_behind_now(const hipfact_handle* h) {
  return h->l_prefix > 0 && h->sp_folded && h->fused_solve && !h->no_dataflow && h->debug_phases == 15 &&
         h->ftop_level < h->plan.nlevels && !h->spanel_side;
}

// queue the numeric factorisation on the stream (values already in d_Kval)
static int factor_enqueue(hipfact_handle* h) {
  const Plan& P = h->plan;
  hipStream_t st = h->stream;
  // the dependency counters of k_factor_top live behind the arena: one fill clears both
  const size_t fill_all = (size_t)P.L_size * sizeof(double) + (size_t)3 * P.nsuper * sizeof(int);
  // the prefix the previous factorisation's solve-panel items zeroed behind them is skipped
  const bool behind = zero_behind_now(h);
  const size_t fill_skip = (behind && h->L_clean) ? (size_t)h->l_prefix * sizeof(double) : 0;
  const size_t fill_bytes = fill_all - fill_skip;
  const bool fill_rides = P.saddle && P.n > 0 && P.m > 0;  // inside k_row_scale (with the info words)
  if (!fill_rides) {
    HCHECK(h, hipMemsetAsync(h->d_info.p, 0, INFO_BYTES, st));
    if (P.L_size > 0) {
      prof_begin(h, PC_MEMSET);
      HCHECK(h, hipMemsetAsync(h->d_L.as<char>() + fill_skip, 0, fill_bytes, st));
      prof_end(h);
    }
  }
  const long long nM = (long long)P.Mi.size();
  const int* vmap = h->maps_on ? h->d_vmap.as<int>() : nullptr;
  double* kprod = h->maps_on ? h->d_Kprod.as<double>() : h->d_Ksc.as<double>();
  if (P.saddle && P.n > 0) {
    // row equilibration (exact powers of two), A^ in pivot order for the solves' SpMVs, scaled
    // copy of K's values for the Schur-complement products and the x update
    if (P.m > 0) {
      const long long nz16 = (long long)((fill_bytes + 15) / 16);
      const int nbz = (int)std::min<long long>(2048, std::max<long long>(1, nz16 / (FB * 8)));
      LAUNCH(PC_GATHER, k_row_scale, dim3(nbz + nblocks((long long)P.m * 16)), dim3(FB), 0, P.m,
             h->d_Ar_ptr.as<int>(), h->d_Ar_col.as<int>(), h->d_Ar_src.as<int>(), h->d_Kval.as<double>(), vmap,
             h->nd > 0 ? h->d_dmask.as<int>() : nullptr, h->equilibrate ? 1 : 0, h->d_dscale.as<double>(),
             h->d_Ar_val.as<double>(), h->nd > 0 ? h->d_Ar_full.as<double>() : nullptr, h->d_Ksc.as<double>(), kprod,
             nbz, reinterpret_cast<double2*>(h->d_L.as<char>() + fill_skip), nz16, h->d_info.as<int>());
    }
  }
  if (nM > 0) {
    if (P.saddle) {
      const long long na = 0;
      const int nbg = 0;
#define MVALS_LAUNCH(IDX, PK)                                                                                       \
  LAUNCH(PC_MVALS, (k_mvals_prod<IDX, PK>), dim3(nblocks(nM, 1 << 16) + nbg), dim3(FB), 0, nM, h->d_prod_ptr.as<IDX>(), \
         h->d_prod_a.as<int>(), h->d_prod_b.as<int>(), h->d_Mtarget.as<IDX>(), kprod,                                  \
         h->d_L.as<double>(), na, nbg, h->d_Ar_src.as<int>(), h->d_Ar_val.as<double>())
      if (h->idx32 && h->prod_packed)
        MVALS_LAUNCH(unsigned int, true);
      else if (h->idx32)
        MVALS_LAUNCH(unsigned int, false);
      else if (h->prod_packed)
        MVALS_LAUNCH(long long, true);
      else
        MVALS_LAUNCH(long long, false);
#undef MVALS_LAUNCH
    } else {
      LAUNCH(PC_MVALS, k_mvals_src, dim3(nblocks(nM, 1 << 16)), dim3(FB), 0, nM, h->d_src.as<int>(),
                         h->d_Mtarget.as<long long>(), h->d_Kval.as<double>(), h->d_L.as<double>());
    }
  }
  if (P.saddle && h->maps_on && P.m > 0)
    LAUNCH(PC_GATHER, k_diag_inactive, dim3(nblocks(P.m)), dim3(FB), 0, P.m, h->d_perm.as<int>(), h->d_cmap.as<int>(),
           h->d_diag_target.as<long long>(), h->d_L.as<double>());
  const int lsplit = (h->debug_phases == 15 && !h->no_dataflow) ? std::min(h->ftop_level, P.nlevels) : P.nlevels;
  if (h->mini_x_bytes > 0 && h->debug_phases == 15 && !h->no_dataflow)
    HCHECK(h, hipMemsetAsync(h->d_xarena.as<double>() + h->mini_x_off, 0xFF, h->mini_x_bytes, st));
  for (int l = 0; l < lsplit; ++l) {
    const LevelInfo& li = h->levels[l];
    const int* it = h->d_items.as<int>();
    const int pull = (li.pull && h->debug_phases == 15) ? 1 : 0;
    if (li.nA > 0 && (h->debug_phases & 1) && !pull)
      LAUNCH(PC_FACTOR_A, k_front_assemble, dim3(li.nA), dim3(1024), li.lds_asm, h->d_sn.as<SnDesc>(), it + li.itA, li.nparts,
             h->d_L.as<double>(), h->d_U.as<double>(), h->d_rel.as<int>(), h->d_child.as<int>());
    if (li.split && h->debug_phases == 15 && li.mini_cnt > 0 && !h->no_dataflow && h->ftop_count > 0) {
      // dense chain: pivot block and panel of the level's one front in ONE small dataflow launch (the panel
      // workgroups follow the posted pivot block tile by tile), then its Schur items at three workgroups per CU
      const FrontItem* fit = h->d_fitems.as<FrontItem>();
      int* flm = reinterpret_cast<int*>(h->d_L.as<double>() + P.L_size);
      LAUNCH(PC_FACTOR_B, k_factor_top, dim3(li.mini_cnt), dim3(512), li.mini_lds,
             h->d_tfitems.as<TopFItem>() + li.mini_off, h->d_L.as<double>(), h->d_U.as<double>(), h->d_info.as<int>(),
             h->d_inv.as<int>(), h->d_rel.as<int>(), flm, flm + P.nsuper, flm + 2 * P.nsuper, h->d_xarena.as<double>(),
             nullptr, nullptr, nullptr, 0);
      if (li.nD > 0)
        LAUNCH(PC_FACTOR_D, k_front_schur<false>, dim3(li.nD), dim3(FB), li.lds_schur, fit + li.itD, h->d_L.as<double>(),
               h->d_U.as<double>(), h->d_inv.as<int>(), h->d_rel.as<int>(), h->d_pullx.as<PullDesc>(), pull);
    } else if (li.split && h->debug_phases == 15) {
      const FrontItem* fit = h->d_fitems.as<FrontItem>();
#define SPLIT_LAUNCHES(CH)                                                                                              \
  LAUNCH(PC_FACTOR_B, k_front_pivot<CH>, dim3(li.count), dim3(512), li.lds_pivot, fit + li.itB, h->d_L.as<double>(),   \
         h->d_U.as<double>(), h->d_info.as<int>(), h->d_inv.as<int>(), h->d_rel.as<int>(), h->d_pullx.as<PullDesc>(), \
         pull);                                                                                                        \
  if (li.nC > 0)                                                                                                       \
    LAUNCH(PC_FACTOR_C, k_front_panel<CH>, dim3(li.nC), dim3(li.panel_threads), li.lds_panel, fit + li.itC,            \
           h->d_L.as<double>(), h->d_U.as<double>(), h->d_inv.as<int>(), h->d_rel.as<int>(),                          \
           h->d_pullx.as<PullDesc>(), pull);                                                                           \
  if (li.nD > 0)                                                                                                       \
    LAUNCH(PC_FACTOR_D, k_front_schur<CH>, dim3(li.nD), dim3(FB), li.lds_schur, fit + li.itD, h->d_L.as<double>(),     \
           h->d_U.as<double>(), h->d_inv.as<int>(), h->d_rel.as<int>(), h->d_pullx.as<PullDesc>(), pull);
      if (li.chain) {
        SPLIT_LAUNCHES(true)
      } else {
        SPLIT_LAUNCHES(false)
      }
#undef SPLIT_LAUNCHES
    } else {
      LAUNCH(PC_FACTOR, k_factor_level, dim3(li.count), dim3(FB), li.lds_factor, h->d_sn.as<SnDesc>(),
             h->d_level_sn.as<int>() + li.begin, h->d_L.as<double>(), h->d_U.as<double>(), h->d_rel.as<int>(),
             h->d_child.as<int>(), h->d_info.as<int>(), h->debug_phases);
    }
  }
  // Solve panels (fused solve): the fronts below the single-launch top of the tree are final here, and that
  // launch is bound by the tree's critical path with most of the chip idle - their panels (most of the bytes)
  // are built beside it on a second stream; the fronts of the top levels follow behind it.
  int sp_done = 0;
  if (h->fused_solve && !h->no_dataflow && h->spanel_side && lsplit < P.nlevels && lsplit > 0 && !h->prof.on && h->side &&
      h->n_sitems == P.nsuper) {
    sp_done = P.level_ptr[lsplit];
    HCHECK(h, hipEventRecord(h->ev_fork, st));
    HCHECK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    hipLaunchKernelGGL(k_build_solve_panels, dim3(sp_done), dim3(SPB), h->sp_lds, h->side, h->d_sitems.as<SolveItem>(),
                       h->d_L.as<double>(), h->d_SPf.as<double>(), h->d_SPb.as<double>());
    HCHECK(h, hipEventRecord(h->ev_join, h->side));
  }
  if (lsplit < P.nlevels) {
    int* fl = reinterpret_cast<int*>(h->d_L.as<double>() + P.L_size);  // cleared with the L arena
    LAUNCH(PC_FACTOR_T, k_factor_top, dim3(h->ftop_count), dim3(512), h->ftop_lds, h->d_tfitems.as<TopFItem>(),
           h->d_L.as<double>(), h->d_U.as<double>(), h->d_info.as<int>(), h->d_inv.as<int>(), h->d_rel.as<int>(), fl,
           fl + P.nsuper, fl + 2 * P.nsuper, h->d_xarena.as<double>(),
           h->sp_folded ? h->d_sitems.as<SolveItem>() : nullptr, h->sp_folded ? h->d_SPf.as<double>() : nullptr,
           h->sp_folded ? h->d_SPb.as<double>() : nullptr, behind ? 1 : 0);
  }
  h->L_clean = behind;  // (whatever else ran leaves the factor in the arena)
  if (h->fused_solve && !h->no_dataflow && !(h->sp_folded && lsplit < P.nlevels)) {
    if (sp_done > 0) HCHECK(h, hipStreamWaitEvent(st, h->ev_join, 0));
    if (h->n_sitems > sp_done)
      LAUNCH(PC_SPANEL, k_build_solve_panels, dim3(h->n_sitems - sp_done), dim3(SPB), h->sp_lds,
             h->d_sitems.as<SolveItem>() + sp_done, h->d_L.as<double>(), h->d_SPf.as<double>(), h->d_SPb.as<double>());
  }
  HCHECK(h, hipGetLastError());
  return dense_setup_async(h);
}

template <class F>
static int run_cached(hipfact_handle* h, int kind, const void* b, void* z, F enqueue, int passes = 0);

static void flush_decide(hipfact_handle* h);
static DecideIn decide_in(hipfact_handle* h);

static int factor_async(hipfact_handle* h) {
  flush_decide(h);  // a deferred verdict is judged against the pivot range of the factorisation it belongs to
  // (two variants of the captured sequence: with the whole zero fill, and without the prefix the previous
  // factorisation left clean)
  const int rc = run_cached(h, 0, nullptr, nullptr, [&] { return factor_enqueue(h); },
                            (zero_behind_now(h) && h->L_clean) ? 1 : 0);
  if (rc) return rc;
  h->num_factor++;
  h->factored = true;
  h->factor_checked = false;
  h->solved = false;
  h->ctl_pending = false;
  // a plan whose previous factorisation needed no correction pass starts without one in its solve graphs (an SQP run
  // refactors the same pattern with slowly changing values); the first solve is checked as always, and a solve that
  // does need a pass is continued at the next synchronising entry point, which also puts the pass back
  h->refine_inline = (h->wc_hint && h->refine_adaptive) ? 0 : h->refine_steps;
  h->inline_probe = true;
  h->seq_at_factor = h->solve_seq;
  h->solves_since_check = 0;
  return HIPFACT_OK;
}

// The dataflow launches exchange data through sentinel-initialised slots and counters; after a
// timed-out launch that state is undefined.  Put all of it back (fresh launches, same process).
static int reset_dataflow_state(hipfact_handle* h) {
  const Plan& P = h->plan;
  hipStream_t st = h->stream;
  if (h->d_xarena.p) HCHECK(h, hipMemsetAsync(h->d_xarena.p, 0xFF, h->d_xarena.bytes, st));
  if (h->d_uvec.p) HCHECK(h, hipMemsetAsync(h->d_uvec.p, 0xFF, std::max<size_t>((size_t)P.u_size * sizeof(double), 16), st));
  if (h->d_ysol.p) HCHECK(h, hipMemsetAsync(h->d_ysol.p, 0xFF, std::max<size_t>((size_t)2 * P.m * sizeof(double), 16), st));
  if (h->d_xhat.p) HCHECK(h, hipMemsetAsync(h->d_xhat.p, 0xFF, std::max<size_t>((size_t)P.m * sizeof(double), 16), st));
  if (h->d_wpart.p) HCHECK(h, hipMemsetAsync(h->d_wpart.p, 0xFF, h->d_wpart.bytes, st));
  if (h->d_spart.p) HCHECK(h, hipMemsetAsync(h->d_spart.p, 0xFF, h->d_spart.bytes, st));
  if (h->d_flags.p) HCHECK(h, hipMemsetAsync(h->d_flags.p, 0, (size_t)4 * P.nsuper * sizeof(int), st));
  HCHECK(h, hipMemsetAsync(h->d_info.p, 0, INFO_BYTES, st));
  HCHECK(h, hipMemsetAsync(h->d_ctl.p, 0, sizeof(RefineCtl), st));
  HCHECK(h, hipStreamSynchronize(st));
  memset(h->h_ctl.p, 0, sizeof(RefineCtl));
  h->solve_seq = h->seq_at_factor = 0;
  h->ctl_pending = false;
  return HIPFACT_OK;
}

// Reads back the info words of the last factorisation (blocking).  Zero / non-finite pivots, a
// negative pivot of the Schur complement A A^T (saddle mode: it is SPD unless the working set is
// rank deficient) and dependency-wait timeouts all invalidate the factorisation.
static int check_info(hipfact_handle* h, const char* phase = "factorisation") {
  HCHECK(h, hipMemcpyAsync(h->h_info.p, h->d_info.p, INFO_WORDS * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  memcpy(h->info_host, h->h_info.p, INFO_WORDS * sizeof(int));
  // The factorisation and the solve kernels count their timed-out waits in the same word.  After an asynchronous
  // refactorisation (hipfact_refactor_device) the first reader may be a solve-phase caller: unless the factorisation
  // has been looked at before, a timeout seen now may be its own and the factor cannot be trusted.
  const bool factor_was_checked = h->factor_checked;
  h->factor_checked = true;
  if (h->fake_timeouts > 0) {
    --h->fake_timeouts;
    h->info_host[INFO_TIMEOUT] += 1;
  }
  char buf[200];
  if (h->info_host[INFO_TIMEOUT] != 0) {
    snprintf(buf, sizeof buf, "dependency wait timed out inside the single-launch %s kernels (%d waits)", phase,
             h->info_host[INFO_TIMEOUT]);
    const bool in_solve = !strcmp(phase, "solve") && factor_was_checked;
    if (!in_solve) h->factored = false;  // a solve does not touch the factor
    h->solved = false;
    const int rc = reset_dataflow_state(h);
    h->error = buf;
    if (!h->no_dataflow) {
      // from now on: per-level launches only (captured graphs hold the dataflow launches: drop them all)
      h->no_dataflow = true;
      h->dataflow_fallbacks++;
      h->graphs.clear();
      for (auto& st2 : h->cache) st2->graphs.clear();
    }
    return rc ? rc : HIPFACT_EINTERNAL;
  }
  if (h->info_host[INFO_ZERO_PIVOT] > 0) {
    snprintf(buf, sizeof buf, "matrix is singular: %d zero or non-finite pivot(s)", h->info_host[INFO_ZERO_PIVOT]);
    h->error = buf;
    h->factored = false;
    return HIPFACT_ESINGULAR;
  }
  if (h->plan.saddle && h->info_host[INFO_NEG_PIVOT] > 0) {
    snprintf(buf, sizeof buf, "working set is numerically rank deficient: %d negative pivot(s) of A A^T",
             h->info_host[INFO_NEG_PIVOT]);
    h->error = buf;
    h->factored = false;
    return HIPFACT_ESINGULAR;
  }
  return HIPFACT_OK;
}

static inline SaddleMaps saddle_maps(const hipfact_handle* h) {
  SaddleMaps M;
  M.vmap = h->maps_on ? h->d_vmap.as<int>() : nullptr;
  M.cmap = h->maps_on ? h->d_cmap.as<int>() : nullptr;
  M.dscale = h->d_dscale.as<double>();
  M.n = h->plan.n;
  return M;
}

// M y = t on the device (y in: t in pivot order, out: solution); skip: device flag that turns
// every launch into a no-op (correction passes of a solve that has already converged)
static void solve_m_async(hipfact_handle* h, const int* skip, const RhsIn* rhs = nullptr, const XupdIn* xup = nullptr) {
  const Plan& P = h->plan;
  if (h->fused_solve && !h->no_dataflow) {
    // (one workgroup more than items: it delivers the deferred verdict of the previous solve, if any; behind it the
    // workgroups of the x update, when it rides in this launch)
    XupdIn X;
    memset(&X, 0, sizeof(X));
    if (xup) X = *xup;
    LAUNCH(PC_TREE, k_solve_tree, dim3(2 * h->n_sitems + 1 + X.nblocks), dim3(ST), 0, h->d_sitems.as<SolveItem>(), h->n_sitems,
           h->d_SPf.as<double>(), h->d_SPb.as<double>(), h->d_sxuoff.as<long long>(), h->d_sxinvoff.as<int>(),
           h->d_inv.as<int>(), h->d_rows.as<int>(), h->d_y.as<double>(),
           h->d_xhat.as<double>(), h->d_uvec.as<double>(), h->d_ysol.as<double>(), P.m, h->d_epoch.as<int>(),
           h->d_info.as<int>(), skip, rhs ? *rhs : RhsIn{nullptr, nullptr, nullptr, nullptr, SaddleMaps{nullptr, nullptr, nullptr, 0}, nullptr},
           decide_in(h), h->d_spart.as<double>(), X);
    return;
  }
  const int ltop = h->no_dataflow ? P.nlevels : std::min(h->top_level, P.nlevels);
  for (int l = 0; l < ltop; ++l) {
    const LevelInfo& li = h->levels[l];
    LAUNCH(PC_FWD, k_fwd_level, dim3(li.count), dim3(SB), li.lds_fwd, h->d_sn.as<SnDesc>(),
           h->d_level_sn.as<int>() + li.begin, h->d_L.as<double>(), h->d_rel.as<int>(), h->d_child.as<int>(),
           h->d_y.as<double>(), h->d_uvec.as<double>(), skip);
  }
  if (ltop < P.nlevels) {
    // two flag sets, one per sweep; each kernel clears the other one's (zero after upload_plan)
    int* ffl = h->d_flags.as<int>();
    int* bfl = ffl + 2 * P.nsuper;
    LAUNCH(PC_FWD, k_fwd_top, dim3(h->top_count), dim3(SB), h->top_lds_fwd, h->d_sn.as<SnDesc>(),
           h->d_titems.as<TopItem>(), ltop, h->d_L.as<double>(), h->d_rel.as<int>(), h->d_child.as<int>(),
           h->d_inv.as<int>(), h->d_ftarget.as<int>(), h->d_y.as<double>(), h->d_uvec.as<double>(),
           ffl, ffl + P.nsuper, h->d_info.as<int>(), bfl, 2 * P.nsuper, h->d_ysol.as<double>(), skip);
    LAUNCH(PC_BWD, k_bwd_top, dim3(h->top_count), dim3(SB), h->top_lds_bwd, h->d_sn.as<SnDesc>(),
           h->d_titems.as<TopItem>(), h->d_L.as<double>(), h->d_rows.as<int>(), h->d_y.as<double>(),
           h->d_wpart.as<double>(), bfl, bfl + P.nsuper, h->d_info.as<int>(), ffl, 2 * P.nsuper,
           h->d_ysol.as<double>(), h->d_uvec.as<double>(), skip);
  }
  for (int l = ltop - 1; l >= 0; --l) {
    const LevelInfo& li = h->levels[l];
    LAUNCH(PC_BWD, k_bwd_level, dim3(li.count), dim3(SB), li.lds_bwd, h->d_sn.as<SnDesc>(),
           h->d_level_sn.as<int>() + li.begin, h->d_L.as<double>(), h->d_rows.as<int>(), h->d_y.as<double>(), skip);
  }
}

// z = K^-1 b (acc: z += K^-1 b) without refinement; b, z device vectors in the caller's
// numbering, b != z
static void solve_once_async(hipfact_handle* h, const double* b, double* z, bool acc, const int* skip, bool raw = false) {
  const Plan& P = h->plan;
  if (h->N_ext == 0) return;
  if (h->nd > 0 && !raw) {
    // dense columns: K_0^-1 b into the scratch vector, then the rank-2k correction writes (or adds) the result
    solve_once_async(h, b, h->d_dtmp.as<double>(), false, skip, true);
    dense_correct_async(h, z, acc, skip);
    return;
  }
  // fused solve launch: the kernel behind it advances the epoch of its double-buffered exchange slots
  int* epoch = (h->fused_solve && !h->no_dataflow && P.m > 0) ? h->d_epoch.as<int>() : nullptr;
  if (P.saddle) {
    const SaddleMaps M = saddle_maps(h);
    if (P.m > 0) {
      if (h->fused_solve && !h->no_dataflow && h->rhs_fused && h->xupd_fused && P.n > 0) {
        // ... and its last workgroups the back substitution of the leaf columns: the whole solve is ONE launch
        const RhsIn R{h->d_Ar_ptr.as<int>(), h->d_Ar_col.as<int>(), h->d_Ar_val.as<double>(), h->d_perm.as<int>(), M, b};
        XupdIn X;
        memset(&X, 0, sizeof(X));
        X.n = P.n;
        X.Kp = h->d_Kp.as<int>();
        X.Ksc = h->d_Ksc.as<double>();
        X.Kc_y = h->d_Kc_y.as<int>();
        X.perm = h->d_perm.as<int>();
        X.M = M;
        X.b = b;
        X.z = z;
        X.acc = acc ? 1 : 0;
        X.nblocks = std::max(1, std::min(h->xupd_blocks, (P.n + (ST / 8) - 1) / (ST / 8)));
        if (h->x_dot_out && !acc) {  // a CG iteration wants the partials of r.g = b_x . z_x (krylov_device.inc)
          X.dot_out = h->x_dot_out;
          h->x_dot_blocks = X.nblocks;
        }
        solve_m_async(h, skip, &R, &X);
        return;
      }
      if (h->fused_solve && !h->no_dataflow && h->rhs_fused) {
        // the forward items of the single launch form their own rows of the right-hand side
        const RhsIn R{h->d_Ar_ptr.as<int>(), h->d_Ar_col.as<int>(), h->d_Ar_val.as<double>(), h->d_perm.as<int>(), M, b};
        solve_m_async(h, skip, &R);
      } else {
        LAUNCH(PC_RHS, k_rhs_saddle, dim3(nblocks((long long)P.m * 16)), dim3(FB), 0, P.m, h->d_Ar_ptr.as<int>(),
               h->d_Ar_col.as<int>(), h->d_Ar_val.as<double>(), h->d_perm.as<int>(), M, b, h->d_y.as<double>(), skip);
        solve_m_async(h, skip);
      }
    }
    if (acc)
      LAUNCH(PC_XUPD, k_x_saddle<true>, dim3(nblocks((long long)P.n * 8)), dim3(FB), 0, P.n, P.m, h->d_Kp.as<int>(),
             h->d_Ksc.as<double>(), h->d_Kc_y.as<int>(), h->d_perm.as<int>(), M, h->d_y.as<double>(), b, z, skip, epoch);
    else
      LAUNCH(PC_XUPD, k_x_saddle<false>, dim3(nblocks((long long)P.n * 8)), dim3(FB), 0, P.n, P.m, h->d_Kp.as<int>(),
             h->d_Ksc.as<double>(), h->d_Kc_y.as<int>(), h->d_perm.as<int>(), M, h->d_y.as<double>(), b, z, skip, epoch);
  } else {
    LAUNCH(PC_PERM, k_gather_skip, dim3(nblocks(P.m)), dim3(FB), 0, (long long)P.m, h->d_perm.as<int>(), b,
           h->d_y.as<double>(), skip);
    solve_m_async(h, skip);
    if (acc)
      LAUNCH(PC_PERM, k_scatter_acc, dim3(nblocks(P.m)), dim3(FB), 0, (long long)P.m, h->d_perm.as<int>(),
             h->d_y.as<double>(), z, skip, epoch);
    else
      LAUNCH(PC_PERM, k_scatter, dim3(nblocks(P.m)), dim3(FB), 0, (long long)P.m, h->d_perm.as<int>(),
             h->d_y.as<double>(), z, epoch);
  }
}

// grid of the residual kernels (= number of partial maxima they leave)
static inline int resid_blocks(const Plan& P) {  // saddle: an even number >= 2 (columns of K | rows of A)
  return P.saddle ? std::max(2, nblocks((long long)P.N * 8, 2048) & ~1) : nblocks(P.N, 2048);
}

static DecideIn decide_in(hipfact_handle* h) {
  // non-adaptive mode (negative target): every in-graph pass runs
  return DecideIn{h->d_ctl.as<RefineCtl>(), static_cast<RefineCtl*>(h->h_ctl_dev), h->d_norms.as<double>(),
                  resid_blocks(h->plan), h->refine_adaptive ? h->refine_tol : -1.0, minmax_ptr(h)};
}

// res = b - K z; updates the refinement control block (first: the residual of the first pass).  defer: no verdict
// launch behind it - the next tree launch (or flush_decide) delivers it.
static void residual_async(hipfact_handle* h, const double* b, const double* z, double* res, bool first,
                           bool defer = false) {
  const Plan& P = h->plan;
  RefineCtl* ctl = h->d_ctl.as<RefineCtl>();
  int* dflag = defer ? &ctl->pending : nullptr;
  if (P.saddle) {
    LAUNCH(PC_RESID, k_residual_saddle, dim3(resid_blocks(P)), dim3(FB), 0, P.n, P.m, h->d_Kp.as<int>(),
           h->d_Ki.as<int>(), h->d_Kval.as<double>(), h->d_Ar_ptr.as<int>(), h->d_Ar_col.as<int>(),
           (h->nd > 0 ? h->d_Ar_full : h->d_Ar_val).as<double>(), h->d_perm.as<int>(), saddle_maps(h), b, z, res, ctl,
           h->d_norms.as<double>(),
           first ? 1 : 0, dflag);
  } else {
    LAUNCH(PC_RESID, k_residual_sym, dim3(resid_blocks(P)), dim3(FB), 0, P.N, h->d_Kp.as<int>(), h->d_Ki.as<int>(),
           h->d_Kval.as<double>(), h->d_Tp.as<int>(), h->d_Ti.as<int>(), h->d_Tsrc.as<int>(), b, z, res, ctl,
           h->d_norms.as<double>(), first ? 1 : 0, dflag);
  }
  if (!defer) LAUNCH(PC_RESID, k_refine_decide, dim3(1), dim3(FB), 0, decide_in(h), first ? 1 : 0, 0);
}

// the verdict of a solve whose graph left it to the next tree launch, for whoever needs it before that
// (a synchronising entry point, a refactorisation - the pivot range it is judged against changes -, a solve
// that does not go through the tree launch); a no-op on the device if it has been delivered already
static void flush_decide(hipfact_handle* h) {
  if (!h->decide_deferred) return;
  hipLaunchKernelGGL(k_refine_decide, dim3(1), dim3(FB), 0, h->stream, decide_in(h), 1, 1);
  h->decide_deferred = false;
}

// (the parked states too: their graphs captured option values - refine_tol, equilibrate, launch variants - by value)
static void drop_graphs(hipfact_handle* h) {
  h->graphs.clear();
  for (auto& st : h->cache) st->graphs.clear();
}
// the solve graphs of the active state only (they capture the length of the caller's vectors)
static void drop_solve_graphs(hipfact_handle* h) {
  std::vector<GraphEntry> keep;
  for (auto& g : h->graphs.v) {
    if (g.kind == 0)
      keep.push_back(g);
    else
      (void)hipGraphExecDestroy(g.exec);
  }
  h->graphs.v.swap(keep);
}

// options that change the plan or the schedule: every cached state is stale
static void invalidate_plans(hipfact_handle* h) {
  h->graphs.clear();
  h->have_plan = false;
  h->factored = false;
  h->cache.clear();
}

// Runs `enqueue` (a function that only queues work on h->stream) through a
// cached hipGraph; falls back to direct enqueueing when graphs are disabled,
// while profiling with events, or when capture is not possible.
template <class F>
static int run_cached(hipfact_handle* h, int kind, const void* b, void* z, F enqueue, int passes) {
  if (!h->use_graph || h->prof.on || h->debug_phases != 15) return enqueue();
  for (auto& g : h->graphs)
    if (g.kind == kind && g.b == b && g.z == z && g.passes == passes) {
      HCHECK(h, hipGraphLaunch(g.exec, h->stream));
      return HIPFACT_OK;
    }
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return enqueue();
  }
  const int rc = enqueue();
  const hipError_t e = hipStreamEndCapture(h->stream, &graph);
  if (rc != HIPFACT_OK || e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (rc != HIPFACT_OK) return rc;
    h->use_graph = false;  // capture unsupported here: stay on the direct path
    return enqueue();
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) {
    (void)hipGetLastError();
    h->use_graph = false;
    return enqueue();
  }
  if (h->graphs.size() >= 16) drop_graphs(h);
  h->graphs.push_back({kind, b, z, passes, exec});
  HCHECK(h, hipGraphLaunch(exec, h->stream));
  return HIPFACT_OK;
}

// `passes` correction passes z += K^-1 res, res and the control block recomputed after each; every
// kernel returns at once when the control block says "done"
static int correct_enqueue(hipfact_handle* h, const double* bb, double* z, int passes) {
  const int* skip = &h->d_ctl.as<RefineCtl>()->done;
  for (int it = 0; it < passes; ++it) {
    solve_once_async(h, h->d_res.as<double>(), z, true, skip);
    residual_async(h, bb, z, h->d_res.as<double>(), false);
  }
  HCHECK(h, hipGetLastError());
  return HIPFACT_OK;
}

// A solve without correction passes in its graph leaves its verdict to the tree launch of the NEXT solve (a
// workgroup of that launch instead of a one-block launch and its kernel boundary behind every solve).
static bool defers_decide(const hipfact_handle* h) {
  return h->decide_lazy && h->refine_steps > 0 && h->refine_adaptive && h->refine_inline == 0 && h->fused_solve &&
         !h->no_dataflow && h->plan.m > 0 && h->plan.saddle;
}

// first pass z = K^-1 b, residual, and the in-graph correction passes
static int solve_enqueue(hipfact_handle* h, const double* b, double* z) {
  const Plan& P = h->plan;
  const double* bb = b;
  if (h->refine_steps > 0 && b == z) {  // keep a private copy of b for the residual
    HCHECK(h, hipMemcpyAsync(h->d_rhs.p, b, (size_t)h->N_ext * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    bb = h->d_rhs.as<double>();
  }
  solve_once_async(h, bb, z, false, nullptr);
  if (h->refine_steps > 0 && !h->skip_resid_now) {
    residual_async(h, bb, z, h->d_res.as<double>(), true, defers_decide(h));
    return correct_enqueue(h, bb, z, h->refine_inline);
  }
  HCHECK(h, hipGetLastError());
  return HIPFACT_OK;
}

// Queues a full solve.  No host synchronisation: the refinement loop is controlled on the device.
static int solve_async(hipfact_handle* h, const double* b, double* z) {
  if (h->N_ext == 0) return HIPFACT_OK;
  if (h->refine_steps > 0 && h->refine_adaptive && h->inline_probe && h->solve_seq > h->seq_at_factor) {
    // Has the previous solve of this factorisation been judged yet?  (A peek at the pinned copy, no
    // synchronisation.)  If its first pass met the tolerance with room to spare, the following
    // solves drop the correction pass from their graph - its kernels would return at once, but 883
    // workgroups per launch still have to be dispatched for that.  A solve that does need a pass is
    // continued at the next synchronising entry point, which also puts the pass back.
    const RefineCtl* hc = h->h_ctl.as<RefineCtl>();
    if (__atomic_load_n(&hc->seq, __ATOMIC_ACQUIRE) == h->solve_seq) {
      if (hc->done && hc->status == 0 && hc->iters == 0 && hc->omega <= 0.25 * hc->tol) {
        h->refine_inline = 0;
        h->wc_hint = true;
      } else {
        h->wc_hint = false;
        h->refine_inline = std::max(h->refine_inline, std::min(h->refine_steps, 1));
      }
      h->inline_probe = false;
    }
  }
  const bool defer = defers_decide(h);
  // (defer <=> the factorisation has been judged well-conditioned: the first pass alone met a quarter of the tolerance)
  const bool unchecked = defer && (!h->inline_probe || h->wc_hint) && h->refine_check_every > 1 &&
                         (h->solves_since_check % h->refine_check_every) != 0;
  h->solves_since_check = unchecked ? h->solves_since_check + 1 : 1;
  if (!(h->fused_solve && !h->no_dataflow && h->plan.m > 0 && h->plan.saddle)) flush_decide(h);  // no tree launch to deliver it
  h->skip_resid_now = unchecked;
  // (an unchecked steady-state solve is two launches: queued directly - replaying a two-node graph measures 4-5 us
  // slower per solve than the launches themselves)
  int rc = unchecked ? solve_enqueue(h, b, z)
                     : run_cached(h, 1, b, z, [&] { return solve_enqueue(h, b, z); },
                                  h->refine_steps > 0 ? (defer ? -2 : h->refine_inline) : -1);
  h->skip_resid_now = false;
  if (rc) return rc;
  if (!unchecked) h->decide_deferred = defer;  // (the tree launch of this solve has delivered an older one)
  h->num_solve++;
  h->solved = true;
  if (h->refine_steps > 0 && !unchecked) h->solve_seq++;
  h->ctl_pending = h->refine_steps > 0 && (!unchecked || h->ctl_pending);
  h->last_b = (b == z) ? h->d_rhs.as<double>() : b;
  h->last_z = z;
  return HIPFACT_OK;
}

// Called by entry points that synchronise anyway: looks at the control block of the last solve,
static bool vtable_can_retry(const hipfact_handle* h);
static int vtable_retry_exact(hipfact_handle* h);

// continues a refinement that is still running, and reports a solve that stalled far above the
// tolerance (numerically singular working set) or a timed-out dataflow launch.
static int finish_solve(hipfact_handle* h, bool* continued = nullptr) {
  if (continued) *continued = false;
  if (!h->ctl_pending) return HIPFACT_OK;
  flush_decide(h);
  HCHECK(h, hipStreamSynchronize(h->stream));
  RefineCtl c;
  memcpy(&c, h->h_ctl.p, sizeof(c));
  int more = 0;
  while (!c.done && h->refine_adaptive && c.iters < h->refine_max) {
    const int passes = std::min(std::max(h->refine_inline, 1), h->refine_max - c.iters);
    const double* b = h->last_b;
    double* z = h->last_z;
    int rc = run_cached(h, 2, b, z, [&] { return correct_enqueue(h, b, z, passes); }, passes);
    if (rc) return rc;
    HCHECK(h, hipStreamSynchronize(h->stream));
    memcpy(&c, h->h_ctl.p, sizeof(c));
    ++more;
  }
  h->ctl_pending = false;
  h->last_ctl = c;
  if (h->inline_probe && h->refine_adaptive && h->refine_steps > 0) {
    // (the verdict of the first solve of this factorisation, read here instead of at the next solve's peek)
    if (c.done && c.status == 0 && c.iters == 0 && c.omega <= 0.25 * c.tol) {
      h->refine_inline = 0;
      h->wc_hint = true;
    }
    h->inline_probe = false;
  }
  if (continued) *continued = more > 0;
  // A refinement that STALLS on a row-dictionary structure which carries rows outside the working set (unit pivots
  // that take part in the ordering): for a nearly rank-deficient working set the quality of the statically pivoted
  // factor as a preconditioner depends on the pivot order, and the order analysed for the rows of THIS K alone has
  // been seen to converge where the superset's stalls (tests: graded family, nearly parallel rows).  Once per
  // factorisation: the dictionary starts over from this K, and the solve is repeated on the new factor.
  if (h->refine_adaptive && c.status == 1 && c.omega > 16.0 * c.tol && vtable_can_retry(h)) {
    int rc = vtable_retry_exact(h);
    if (rc) return rc;
    return finish_solve(h, continued);
  }
  if (c.iters > 0) h->num_refined++;
  h->num_passes += c.iters;
  // the next solves of this factorisation carry as many passes in their graph as this one needed
  if (more > 0) {
    h->refine_inline = std::min(std::max(h->refine_inline, c.iters), 4);
    h->wc_hint = false;
  }
  if (h->refine_adaptive && c.status != 2 && c.omega > h->fail_omega) {
    char buf[200];
    snprintf(buf, sizeof buf,
             "working set is numerically singular: iterative refinement stalled at backward error %.2e after %d "
             "passes (pivot-ratio condition estimate %.2e)", c.omega, c.iters, c.kappa);
    h->error = buf;
    return HIPFACT_ESINGULAR;
  }
  return HIPFACT_OK;
}

// word-wise FNV-1a (the patterns are megabytes: this runs at memory speed)
static unsigned long long hash_ints(const int* p, size_t n, unsigned long long hsh = 1469598103934665603ull) {
  size_t i = 0;
  for (; i + 1 < n; i += 2) {
    unsigned long long w;
    memcpy(&w, p + i, 8);
    hsh = (hsh ^ w) * 1099511628211ull;
  }
  if (i < n) hsh = (hsh ^ (unsigned long long)(unsigned int)p[i]) * 1099511628211ull;
  return hsh;
}

// parks the active plan state in the LRU list (evicting the least recently used one) and leaves a
// fresh state active
static void park_active(hipfact_handle* h) {
  if (h->have_plan && h->plan_cache_max > 0) {
    if ((int)h->cache.size() >= h->plan_cache_max) {
      size_t lru = 0;
      for (size_t i = 1; i < h->cache.size(); ++i)
        if (h->cache[i]->use_stamp < h->cache[lru]->use_stamp) lru = i;
      h->cache.erase(h->cache.begin() + (long)lru);
    }
    h->cache.emplace_back(new PlanState(std::move(static_cast<PlanState&>(*h))));
  }
  static_cast<PlanState&>(*h) = PlanState();
}

// makes cache[i] the active state (the active one takes its place in the list)
static void swap_in(hipfact_handle* h, size_t i) {
  PlanState tmp(std::move(*h->cache[i]));
  *h->cache[i] = std::move(static_cast<PlanState&>(*h));
  static_cast<PlanState&>(*h) = std::move(tmp);
  if (!h->cache[i]->have_plan) h->cache.erase(h->cache.begin() + (long)i);
  h->plan_swaps++;
}

static int ensure_plan(hipfact_handle* h, int N, const int* colptr, const int* rowidx, const double* vals) {
  const long long nnz = N > 0 ? colptr[N] : 0;
  // the hash only serves the search of the LRU: the active state is compared directly first (steady state of an
  // SQP run: same pattern again), which makes hashing 5 MB per call unnecessary
  unsigned long long hsh = 0;
  bool hashed = false;
  auto matches = [&](const PlanState& s) {
    if (!(s.have_plan && !s.from_jacobian && s.plan.N == N && s.plan.nnzK == nnz && (!hashed || s.key_hash == hsh) &&
          memcmp(s.plan.Kp.data(), colptr, (size_t)(N + 1) * sizeof(int)) == 0 &&
          (nnz == 0 || memcmp(s.plan.Ki.data(), rowidx, (size_t)nnz * sizeof(int)) == 0)))
      return false;
    // pattern unchanged.  The saddle classification also depends on the unit diagonal values (n entries)
    if (s.plan.saddle && vals)
      for (int j = 0; j < s.plan.n; ++j)
        if (vals[colptr[j]] != 1.0) return false;
    // ... both ways: a plan analysed as a general symmetric matrix (its values had no unit diagonal then) is not
    // the plan for values that do make the structure a saddle matrix (K is indefinite: the constrained pivot order
    // of the saddle mode is what makes static pivots safe)
    if (!s.plan.saddle && s.plan.saddle_shape && vals) {
      bool unit = true;
      for (int j = 0; j < s.plan.n_shape && unit; ++j) unit = vals[colptr[j]] == 1.0;
      if (unit) return false;
    }
    return true;
  };
  bool hit = matches(*h);
  if (!hit) {
    hsh = hash_ints(rowidx, (size_t)nnz, hash_ints(colptr, (size_t)N + 1));
    hashed = true;
  }
  for (size_t i = 0; !hit && i < h->cache.size(); ++i)
    if (matches(*h->cache[i])) {
      swap_in(h, i);
      hit = true;
    }
  h->use_stamp = ++h->use_clock;
  if (hit) {
    h->cache_hits++;
    return HIPFACT_OK;
  }
  park_active(h);
  try {
    if (!build_plan(N, colptr, rowidx, vals, h->prm, h->plan)) {
      h->error = h->plan.error;
      return HIPFACT_EINVAL;
    }
  } catch (const std::bad_alloc&) {
    h->error = "out of host memory during analysis";
    return HIPFACT_ENOMEM;
  }
  h->analyses++;
  h->key_hash = hsh;
  h->N_ext = N;
  h->use_stamp = h->use_clock;
  int rc = upload_plan(h);
  if (rc) return rc;
  h->have_plan = true;
  return HIPFACT_OK;
}

// factorisation + verdict; a timed-out dataflow launch is repeated once on the per-level path (fresh launches,
// same process)
static int check_factor(hipfact_handle* h, bool could_fall_back);
static int factor_and_check(hipfact_handle* h) {
  const bool could_fall_back = !h->no_dataflow;
  int rc = factor_async(h);
  if (rc) return rc;
  return check_factor(h, could_fall_back);
}
// the verdict on a queued factorisation (synchronises); a timed-out dataflow launch is repeated on the per-level path
static int check_factor(hipfact_handle* h, bool could_fall_back) {
  int rc = check_info(h);
  if (rc == HIPFACT_EINTERNAL && could_fall_back && h->no_dataflow && h->info_host[INFO_TIMEOUT] != 0) {
    if ((rc = factor_async(h))) return rc;
    rc = check_info(h);
  }
  return rc;
}

// roctx ranges around the boundary calls (set_matrix / solve / solution), switched on with HIPFACT_ROCTX=1: the
// library is looked up at run time (no link dependency; rocprofv3 --marker-trace shows the ranges)
struct RoctxRange {
  typedef int (*push_fn)(const char*);
  typedef int (*pop_fn)();
  static pop_fn& pop_ptr() {
    static pop_fn p = nullptr;
    return p;
  }
  static push_fn push_ptr() {
    static push_fn p = [] {
      push_fn f = nullptr;
      const char* e = getenv("HIPFACT_ROCTX");
      if (e && atoi(e) != 0) {
        void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (lib) {
          f = reinterpret_cast<push_fn>(dlsym(lib, "roctxRangePushA"));
          pop_ptr() = reinterpret_cast<pop_fn>(dlsym(lib, "roctxRangePop"));
          if (!pop_ptr()) f = nullptr;
        }
      }
      return f;
    }();
    return p;
  }
  bool on = false;
  explicit RoctxRange(const char* name) {
    if (push_fn f = push_ptr()) {
      f(name);
      on = true;
    }
  }
  ~RoctxRange() {
    if (on) pop_ptr()();
  }
};

static int enter(hipfact_handle* h) {
  if (!h) return HIPFACT_EINVAL;
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) {
    h->error = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return HIPFACT_EDEVICE;
  }
  return HIPFACT_OK;
}

template <int LANES>
static void launch_spmv(hipStream_t st, int nrows, const int* ptr, const int* idx, const double* val, const int* ptr2,
                        const int* idx2, const double* val2, const double* x, double* y) {
  const int rows_per_block = FB / LANES;
  long long blocks = ((long long)nrows + rows_per_block - 1) / rows_per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_spmv_csr<LANES>, dim3((int)blocks), dim3(FB), 0, st, nrows, ptr, idx, val, ptr2, idx2, val2,
                     x, y);
}

extern "C" {

int hipfact_create(hipfact_handle** out, int device) {
  if (!out) return HIPFACT_EINVAL;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                     "); hipfact has no CPU fallback";
    return HIPFACT_EDEVICE;
  }
  if (device < 0) {
    const char* s = getenv("SLEQP_HIP_DEVICE");
    if (!s) s = getenv("LOCAL_RANK");
    device = s ? atoi(s) : 0;
    if (device < 0 || device >= count) device = device % count;
  }
  if (device >= count) {
    g_create_error = "device ordinal out of range";
    return HIPFACT_EDEVICE;
  }
  hipfact_handle* h = new (std::nothrow) hipfact_handle();
  if (!h) return HIPFACT_ENOMEM;
  h->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming)) != hipSuccess) {
    g_create_error = std::string("device init: ") + hipGetErrorString(e);
    delete h;
    return HIPFACT_EDEVICE;
  }
  if (const char* s = getenv("HIPFACT_REFINE")) h->refine_steps = atoi(s);
  if (const char* s = getenv("HIPFACT_SPLIT_MAX")) h->split_max_fronts = atoi(s);
  if (const char* s = getenv("HIPFACT_PULL_MAX")) h->pull_max_children = atoi(s);
  if (const char* s = getenv("HIPFACT_TOP_PREFETCH")) h->top_prefetch = atoi(s);
  if (const char* s = getenv("HIPFACT_WIDE_MIN")) h->wide_min_rows = atoi(s);
  if (const char* s = getenv("HIPFACT_FACTOR_TOP")) h->factor_top_max = atoi(s);
  if (const char* s = getenv("HIPFACT_SOLVE_SORTED")) h->solve_sorted = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_FACTOR_FINE")) h->factor_top_fine = atoi(s);
  if (const char* s = getenv("HIPFACT_FACTOR_POST")) h->factor_top_post = atoi(s);
  if (const char* s = getenv("HIPFACT_SPANEL_FOLD")) h->spanel_fold = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_ZERO_BEHIND")) h->zero_behind = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_RHS_FUSED")) h->rhs_fused = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_XUPD_FUSED")) h->xupd_fused = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_SOLVE_WHOLE_MAX")) h->solve_whole_max = std::max(SOLVE_PREFETCH, atoi(s));
  if (const char* s = getenv("HIPFACT_CG_GRAPH")) h->cg_graph = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_XUPD_BLOCKS")) h->xupd_blocks = std::max(1, atoi(s));
  if (const char* s = getenv("HIPFACT_SOLVE_SLICES")) h->solve_slices = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_CHAIN_FUSE")) h->chain_fuse = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_DECIDE_LAZY")) h->decide_lazy = atoi(s) != 0;
  if (const char* s = getenv("HIPFACT_SPANEL_ROOM")) h->spanel_fold_room = atoi(s);
  if (const char* s = getenv("HIPFACT_PANEL_SMALL")) h->panel_small_below = atoi(s);
  if (const char* s = getenv("HIPFACT_TOP_MAX")) h->top_max_fronts = atoi(s);
  if (const char* s = getenv("HIPFACT_GRAPH")) h->use_graph = atoi(s) != 0;
  *out = h;
  return HIPFACT_OK;
}

int hipfact_retain(hipfact_handle* h) {
  if (!h) return HIPFACT_EINVAL;
  h->refcount.fetch_add(1);
  return HIPFACT_OK;
}

int hipfact_free(hipfact_handle** handle) {
  if (!handle || !*handle) return HIPFACT_OK;
  hipfact_handle* h = *handle;
  *handle = nullptr;
  if (h->refcount.fetch_sub(1) > 1) return HIPFACT_OK;  // other owners remain
  (void)hipSetDevice(h->device);
  if (h->stream) {
    (void)hipStreamSynchronize(h->stream);
    drop_graphs(h);
    h->cache.clear();
    (void)hipStreamDestroy(h->stream);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  }
  delete h;
  return HIPFACT_OK;
}

const char* hipfact_last_error(const hipfact_handle* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

static int set_matrix_virtual(hipfact_handle* h, int N, const int* kp, const int* ki, const double* kx, bool* handled);

int hipfact_set_matrix(hipfact_handle* h, int N, const int* colptr, const int* rowidx, const double* vals) {
  RoctxRange range("hipfact_set_matrix");
  int rc = enter(h);
  if (rc) return rc;
  if (N < 0 || !colptr || (N > 0 && colptr[N] > 0 && (!rowidx || !vals))) {
    h->error = "hipfact_set_matrix: invalid arguments";
    return HIPFACT_EINVAL;
  }
  // An augmented matrix [I A_W^T; A_W 0] goes through the row dictionary (vtable_superset.inc): a working set made of
  // rows seen before is a numeric refactorisation, whatever it does to the pattern of K
  {
    bool handled = false;
    if ((rc = set_matrix_virtual(h, N, colptr, rowidx, vals, &handled))) return rc;
    if (handled) return HIPFACT_OK;
  }
  // Steady state of an SQP run: the pattern is the active plan's.  All the host has to do then is compare 5 MB of
  // indices - which it can do WHILE the device works: the values and the factorisation are queued first on the
  // assumption that the pattern matches, and the comparison runs beside them.  If it does not match, what was
  // queued is discarded (it ran on the old plan's own buffers) and the ordinary path follows.
  const long long nnz_in = N > 0 ? colptr[N] : 0;
  if (h->speculate && h->have_plan && !h->from_jacobian && h->plan.N == N && h->plan.nnzK == nnz_in &&
      (size_t)nnz_in * sizeof(double) >= (64u << 10)) {
    const int analyses = h->analyses, swaps = h->plan_swaps;
    const bool could_fall_back = !h->no_dataflow;
    HCHECK(h, hipStreamSynchronize(h->stream));
    HCHECK(h, hipMemcpyAsync(h->d_Kval.p, vals, (size_t)nnz_in * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // (the copy reads the caller's array: no return, error or not, while it may be in flight; and the queued work
    // runs on the active state's buffers and graphs: it is awaited before that state is parked or swapped)
    if ((rc = factor_async(h))) {
      (void)hipStreamSynchronize(h->stream);
      return rc;
    }
    const bool same_active = h->plan.N == N && h->plan.nnzK == nnz_in &&
                             memcmp(h->plan.Kp.data(), colptr, (size_t)(N + 1) * sizeof(int)) == 0 &&
                             (nnz_in == 0 || memcmp(h->plan.Ki.data(), rowidx, (size_t)nnz_in * sizeof(int)) == 0);
    if (!same_active) HCHECK(h, hipStreamSynchronize(h->stream));
    if ((rc = ensure_plan(h, N, colptr, rowidx, vals))) {
      (void)hipStreamSynchronize(h->stream);
      return rc;
    }
    if (h->analyses == analyses && h->plan_swaps == swaps) return check_factor(h, could_fall_back);
    HCHECK(h, hipStreamSynchronize(h->stream));  // another plan is active now: start over on it
  } else if ((rc = ensure_plan(h, N, colptr, rowidx, vals))) {
    return rc;
  }
  const size_t nnz = (size_t)h->plan.nnzK;
  if (nnz > 0) {
    HCHECK(h, hipStreamSynchronize(h->stream));  // a copy out of the staging buffer may still be in flight
    // Large arrays go to the copy engine straight from the caller's (pageable) memory: the runtime pins them in
    // place for the duration of the copy, which measures 209 us for 8.8 MB against 343 us through the pinned
    // staging buffer and is as fast as a permanently registered array (scripts/probe/h2d_paths.hip) without
    // holding a registration on memory this library does not own.  check_info below synchronises, so the array
    // has been read before this call returns.
    const void* src = vals;
    const size_t bytes = nnz * sizeof(double);
    if (bytes < (64u << 10)) {
      HCHECK(h, h->h_stage.ensure(bytes));
      memcpy(h->h_stage.p, vals, bytes);
      src = h->h_stage.p;
    }
    HCHECK(h, hipMemcpyAsync(h->d_Kval.p, src, bytes, hipMemcpyHostToDevice, h->stream));
  }
  return factor_and_check(h);
}

int hipfact_refactor_device(hipfact_handle* h, const double* d_vals) {
  int rc = enter(h);
  if (rc) return rc;
  if (!h->have_plan) {
    h->error = "hipfact_refactor_device: no matrix pattern set";
    return HIPFACT_ESTATE;
  }
  if (d_vals && d_vals != h->d_Kval.p && h->plan.nnzK > 0)
    HCHECK(h, hipMemcpyAsync(h->d_Kval.p, d_vals, (size_t)h->plan.nnzK * sizeof(double), hipMemcpyDeviceToDevice,
                             h->stream));
  return factor_async(h);
}

static int require_factor(hipfact_handle* h, const char* who) {
  if (!h->have_plan || !h->factored) {
    h->error = std::string(who) + ": no factorisation (call hipfact_set_matrix first)";
    return HIPFACT_ESTATE;
  }
  return HIPFACT_OK;
}

int hipfact_solve_dense(hipfact_handle* h, const double* rhs) {
  RoctxRange range("hipfact_solve_dense");
  int rc = enter(h);
  if (rc) return rc;
  if ((rc = require_factor(h, "hipfact_solve_dense"))) return rc;
  const size_t N = (size_t)h->N_ext;
  if (N == 0) return HIPFACT_OK;
  if (!rhs) return HIPFACT_EINVAL;
  HCHECK(h, hipStreamSynchronize(h->stream));  // staging buffer may still be in flight
  HCHECK(h, h->h_stage.ensure(N * sizeof(double)));
  memcpy(h->h_stage.p, rhs, N * sizeof(double));
  HCHECK(h, hipMemcpyAsync(h->d_rhs.p, h->h_stage.p, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return solve_async(h, h->d_rhs.as<double>(), h->d_sol.as<double>());
}

int hipfact_solve_sparse(hipfact_handle* h, int dim, int nnz, const int* indices, const double* data) {
  RoctxRange range("hipfact_solve_sparse");
  int rc = enter(h);
  if (rc) return rc;
  if ((rc = require_factor(h, "hipfact_solve_sparse"))) return rc;
  const int N = h->N_ext;
  if (dim != N || nnz < 0 || nnz > N || (nnz > 0 && (!indices || !data))) {
    h->error = "hipfact_solve_sparse: rhs dimension does not match the matrix";
    return HIPFACT_EINVAL;
  }
  if (N == 0) return HIPFACT_OK;
  for (int k = 0; k < nnz; ++k)
    if (indices[k] < 0 || indices[k] >= N) {
      h->error = "hipfact_solve_sparse: index out of range";
      return HIPFACT_EINVAL;
    }
  HCHECK(h, hipMemsetAsync(h->d_rhs.p, 0, (size_t)N * sizeof(double), h->stream));
  bool borrowed = false;
  if (nnz > 0) {
    const size_t bytes = (size_t)nnz * (sizeof(double) + sizeof(int));
    HCHECK(h, h->d_sp_val.ensure((size_t)nnz * sizeof(double)));
    HCHECK(h, h->d_sp_idx.ensure((size_t)nnz * sizeof(int)));
    const void *pv = data, *pi = indices;
    if (bytes >= (64u << 10)) {
      // long vectors straight from the caller's arrays (the copy engine pins them in place, see hipfact_set_matrix);
      // they are only borrowed for this call: the copies are awaited below, the solve is not
      borrowed = true;
    } else {
      HCHECK(h, hipStreamSynchronize(h->stream));
      HCHECK(h, h->h_stage.ensure(bytes));
      double* sv = h->h_stage.as<double>();
      int* si = reinterpret_cast<int*>(sv + nnz);
      memcpy(sv, data, (size_t)nnz * sizeof(double));
      memcpy(si, indices, (size_t)nnz * sizeof(int));
      pv = sv;
      pi = si;
    }
    hipError_t ce = hipMemcpyAsync(h->d_sp_val.p, pv, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (ce == hipSuccess)
      ce = hipMemcpyAsync(h->d_sp_idx.p, pi, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, h->stream);
    if (ce == hipSuccess && borrowed) ce = hipEventRecord(h->ev_fork, h->stream);
    if (ce != hipSuccess) {
      (void)hipStreamSynchronize(h->stream);  // a copy out of the caller's arrays may be in flight
      HCHECK(h, ce);
    }
    hipLaunchKernelGGL(k_scatter_sparse, dim3(nblocks(nnz)), dim3(FB), 0, h->stream, nnz, h->d_sp_idx.as<int>(),
                       h->d_sp_val.as<double>(), h->d_rhs.as<double>());
  }
  rc = solve_async(h, h->d_rhs.as<double>(), h->d_sol.as<double>());
  if (borrowed) {
    const hipError_t se = hipEventSynchronize(h->ev_fork);
    if (se != hipSuccess) (void)hipStreamSynchronize(h->stream);
    if (rc == HIPFACT_OK) HCHECK(h, se);
  }
  return rc;
}

int hipfact_solve_device(hipfact_handle* h, const double* d_rhs, double* d_sol) {
  int rc = enter(h);
  if (rc) return rc;
  if ((rc = require_factor(h, "hipfact_solve_device"))) return rc;
  if (h->N_ext > 0 && (!d_rhs || !d_sol)) return HIPFACT_EINVAL;
  return solve_async(h, d_rhs, d_sol);
}

int hipfact_solution(hipfact_handle* h, double* out, int begin, int end) {
  RoctxRange range("hipfact_solution");
  int rc = enter(h);
  if (rc) return rc;
  if (!h->solved && h->N_ext > 0) {
    h->error = "hipfact_solution: no solve has been performed";
    return HIPFACT_ESTATE;
  }
  if (begin < 0 || end < begin || end > h->N_ext || (end > begin && !out)) {
    h->error = "hipfact_solution: range outside [0, N]";
    return HIPFACT_EINVAL;
  }
  const size_t cnt = (size_t)(end - begin);
  // the refinement of the last solve is finished (and a stalled one reported) before its result leaves
  if ((rc = finish_solve(h))) return rc;
  if (cnt == 0) return HIPFACT_OK;
  HCHECK(h, hipStreamSynchronize(h->stream));
  // long ranges straight into the caller's array (check_info below synchronises)
  const bool direct = cnt * sizeof(double) >= (64u << 10);
  if (!direct) HCHECK(h, h->h_stage.ensure(cnt * sizeof(double)));
  void* dst = direct ? static_cast<void*>(out) : h->h_stage.p;
  HCHECK(h, hipMemcpyAsync(dst, h->d_sol.as<double>() + begin, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  const bool could_fall_back = !h->no_dataflow;
  rc = check_info(h, "solve");  // synchronises
  if (rc == HIPFACT_EINTERNAL && could_fall_back && h->no_dataflow && h->last_b && h->last_z) {
    // the sweep timed out: the same solve once more through the per-level kernels - behind a fresh factorisation
    // when the timed-out wait may have been the (unchecked) factorisation's own
    if (!h->factored) {
      if ((rc = factor_async(h))) return rc;
      if ((rc = check_info(h))) return rc;
    }
    if ((rc = solve_async(h, h->last_b, h->last_z))) return rc;
    if ((rc = finish_solve(h))) return rc;
    HCHECK(h, hipMemcpyAsync(dst, h->d_sol.as<double>() + begin, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    rc = check_info(h, "solve");
  }
  if (rc) return rc;
  if (!direct) memcpy(out, h->h_stage.p, cnt * sizeof(double));
  return HIPFACT_OK;
}

int hipfact_check(hipfact_handle* h) {
  int rc = enter(h);
  if (rc) return rc;
  if (!h->have_plan) return HIPFACT_OK;
  if ((rc = finish_solve(h))) return rc;
  return check_info(h, h->solved ? "solve" : "factorisation");
}

int hipfact_solution_device(hipfact_handle* h, const double** d_sol) {
  if (!h || !d_sol) return HIPFACT_EINVAL;
  *d_sol = h->d_sol.as<double>();
  return HIPFACT_OK;
}

int hipfact_condition(hipfact_handle* h, double* condition) {
  int rc = enter(h);
  if (rc) return rc;
  if ((rc = require_factor(h, "hipfact_condition"))) return rc;
  if (!condition) return HIPFACT_EINVAL;
  const Plan& P = h->plan;
  if (P.nsuper == 0) {
    *condition = 1.0;
    return HIPFACT_OK;
  }
  // min / max |d| were left behind the info words by the factorisation
  unsigned long long* hm = reinterpret_cast<unsigned long long*>(h->h_info.as<char>() + INFO_WORDS * sizeof(int));
  HCHECK(h, hipMemcpyAsync(hm, minmax_ptr(h), 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  const unsigned long long nlo = ~hm[0];
  double lo, hi;
  memcpy(&lo, &nlo, 8);
  memcpy(&hi, &hm[1], 8);
  if (P.saddle) {  // the leaf pivots of the identity block are 1
    lo = std::min(lo, 1.0);
    hi = std::max(hi, 1.0);
  }
  *condition = (lo > 0.0) ? hi / lo : INFINITY;
  return HIPFACT_OK;
}

int hipfact_synchronize(hipfact_handle* h) {
  int rc = enter(h);
  if (rc) return rc;
  HCHECK(h, hipStreamSynchronize(h->stream));
  return HIPFACT_OK;
}

int hipfact_stream(hipfact_handle* h, void** stream) {
  if (!h || !stream) return HIPFACT_EINVAL;
  *stream = (void*)h->stream;
  return HIPFACT_OK;
}

// ---------------------------------------------------------------------------
// Plan for the structure K_s = [I J_s^T; J_s 0] over the constraint rows `cover` (sidx >= 0) of J.
static int build_superset_plan(hipfact_handle* h, int n, int m_total, const int* jp, const int* ji,
                               const std::vector<int>& sidx, int ms) {
  const int jnnz = n > 0 ? jp[n] : 0;
  const int Ns = n + ms;
  std::vector<int> kp((size_t)Ns + 1), ki;
  ki.reserve((size_t)n + jnnz);
  for (int j = 0; j < n; ++j) {
    kp[j] = (int)ki.size();
    ki.push_back(j);
    for (int q = jp[j]; q < jp[j + 1]; ++q)
      if (sidx[ji[q]] >= 0) ki.push_back(n + sidx[ji[q]]);
  }
  for (int j = n; j <= Ns; ++j) kp[j] = (int)ki.size();
  try {
    if (!build_plan(Ns, kp.data(), ki.data(), nullptr, h->prm, h->plan)) {
      h->error = h->plan.error;
      return HIPFACT_EINVAL;
    }
  } catch (const std::bad_alloc&) {
    h->error = "out of host memory during analysis";
    return HIPFACT_ENOMEM;
  }
  if (!h->plan.saddle || h->plan.n != n) {
    h->error = "internal: superset structure not recognised as a saddle matrix";
    return HIPFACT_EINTERNAL;
  }
  h->analyses++;
  h->from_jacobian = true;
  h->Jp.assign(jp, jp + n + 1);
  h->Ji.assign(ji, ji + jnnz);
  h->sidx = sidx;
  h->m_struct = ms;
  h->N_ext = 2 * n + ms;  // capacity of the solve vectors: any working set inside the superset plus all bounds
  int rc = upload_plan(h);
  if (rc) return rc;
  std::vector<int> srow((size_t)std::max(ms, 1), 0);
  for (int i = 0; i < m_total; ++i)
    if (sidx[i] >= 0) srow[sidx[i]] = i;
  std::vector<long long> dt((size_t)std::max(h->plan.m, 1), 0);
  for (int k = 0; k < h->plan.m; ++k) dt[k] = h->plan.Mtarget[h->plan.Mp[k]];  // the diagonal comes first in every column
  if ((rc = upload(h, h->d_sidx, h->sidx))) return rc;
  if ((rc = upload(h, h->d_cmap, srow))) return rc;  // placeholder sizing; d_srow below
  if ((rc = upload(h, h->d_srow, srow))) return rc;
  if ((rc = upload(h, h->d_diag_target, dt))) return rc;
  HCHECK(h, h->d_vmap.ensure(std::max<size_t>((size_t)n * sizeof(int), 16)));
  HCHECK(h, h->d_Kprod.ensure(std::max<size_t>((size_t)h->plan.nnzK * sizeof(double), 16)));
  h->have_plan = true;
  return HIPFACT_OK;
}

// ---------------------------------------------------------------------------
// Superset path of the device assembly: finds (or analyses) a plan whose structure [I J_s^T; J_s 0] covers the working
// set's constraint rows, then a numeric refactorisation.  The Jacobian (d_jp / d_ji / d_jx) and the working-set maps
// (d_vi / d_ci) are on the device already; j_colptr / j_rowidx / cons_index are the host copies.
static int superset_refactor(hipfact_handle* h, int n, int m_total, const int* j_colptr, const int* j_rowidx,
                             const int* cons_index, int nav, int nac, int N, unsigned long long jhash, bool known) {
  int rc;
  const int jnnz = n > 0 ? j_colptr[n] : 0;
  hipStream_t st = h->stream;
  // ---- superset path: find a plan whose structure covers the working set's constraint rows
  auto covers = [&](const PlanState& s) {
    if (!(&s == static_cast<const PlanState*>(h) && known) &&
        !(s.have_plan && s.from_jacobian && s.plan.n == n && (int)s.sidx.size() == m_total && s.key_hash == jhash &&
          (int)s.Ji.size() == jnnz && memcmp(s.Jp.data(), j_colptr, (size_t)(n + 1) * sizeof(int)) == 0 &&
          (jnnz == 0 || memcmp(s.Ji.data(), j_rowidx, (size_t)jnnz * sizeof(int)) == 0)))
      return false;
    for (int i = 0; i < m_total; ++i)
      if (cons_index[i] >= 0 && s.sidx[i] < 0) return false;
    // a structure far larger than the working set wastes the factorisation on unit rows
    return 2LL * nac >= s.m_struct || s.m_struct - nac <= 256;
  };
  bool hit = covers(*h);
  for (size_t i = 0; !hit && i < h->cache.size(); ++i)
    if (covers(*h->cache[i])) {
      swap_in(h, i);
      hit = true;
    }
  h->use_stamp = ++h->use_clock;
  if (hit) {
    h->cache_hits++;
  } else {
    // new structure: every row of J when the working set holds most of them (rows then enter and leave
    // without re-analysis), else the working set's rows plus those of a recent structure of this
    // Jacobian when that stays close (working sets that oscillate)
    std::vector<int> sidx((size_t)m_total, -1);
    std::vector<char> cover((size_t)m_total, 0);
    int ms = 0;
    if (2LL * nac >= m_total) {
      std::fill(cover.begin(), cover.end(), 1);
    } else {
      for (int i = 0; i < m_total; ++i) cover[i] = cons_index[i] >= 0;
      const PlanState* prev = nullptr;
      auto same_j = [&](const PlanState& s) {
        return s.have_plan && s.from_jacobian && s.key_hash == jhash && (int)s.sidx.size() == m_total && s.plan.n == n;
      };
      if (same_j(*h)) prev = h;
      for (size_t i = 0; !prev && i < h->cache.size(); ++i)
        if (same_j(*h->cache[i])) prev = h->cache[i].get();
      if (prev) {
        int uni = 0;
        for (int i = 0; i < m_total; ++i) uni += (cover[i] || prev->sidx[i] >= 0);
        if (uni <= nac + nac / 4 + 64)
          for (int i = 0; i < m_total; ++i) cover[i] = cover[i] || prev->sidx[i] >= 0;
      }
    }
    for (int i = 0; i < m_total; ++i)
      if (cover[i]) sidx[i] = ms++;
    park_active(h);
    h->key_hash = jhash;
    h->use_stamp = h->use_clock;
    if ((rc = build_superset_plan(h, n, m_total, j_colptr, j_rowidx, sidx, ms))) return rc;
  }
  h->maps_on = true;
  // (a solve graph whose right-hand side and solution alias captured a copy of N_ext doubles)
  if (h->N_ext != N) drop_solve_graphs(h);
  h->N_ext = N;
  h->n_bounds = nav;
  const Plan& P = h->plan;
  hipLaunchKernelGGL(k_struct_fill, dim3(nblocks(std::max(n, h->m_struct))), dim3(FB), 0, st, n, h->m_struct,
                     h->d_jp.as<int>(), h->d_ji.as<int>(), h->d_jx.as<double>(), h->d_vi.as<int>(), h->d_ci.as<int>(),
                     h->d_sidx.as<int>(), h->d_srow.as<int>(), h->d_Kp.as<int>(), h->d_Kval.as<double>(),
                     h->d_vmap.as<int>(), h->d_cmap.as<int>());
  HCHECK(h, hipGetLastError());
  (void)P;
  return factor_and_check(h);
}

#include "vtable_superset.inc"

int hipfact_assemble_kkt(hipfact_handle* h, int n, int m_total, const int* j_colptr, const int* j_rowidx,
                         const double* j_vals, const int* var_index, const int* cons_index, int working_set_size,
                         int* k_nnz, int* k_colptr, int* k_rowidx, double* k_vals) {
  int rc = enter(h);
  if (rc) return rc;
  if (n < 0 || m_total < 0 || working_set_size < 0 || !j_colptr || !var_index || (m_total > 0 && !cons_index)) {
    h->error = "hipfact_assemble_kkt: invalid arguments";
    return HIPFACT_EINVAL;
  }
  const int jnnz = n > 0 ? j_colptr[n] : 0;
  const int N = n + working_set_size;
  h->vj->dev_current = false;  // the device copies of the Jacobian and the maps are this call's from here on
  int nav = 0, nac = 0;
  for (int j = 0; j < n; ++j) nav += (var_index[j] >= 0);
  for (int i = 0; i < m_total; ++i) nac += (cons_index[i] >= 0);
  if (nav + nac != working_set_size) {
    h->error = "hipfact_assemble_kkt: working_set_size does not match the index maps";
    return HIPFACT_EINVAL;
  }
  // Steady state of an SQP run: the pattern of J is the one the active plan was built from.  One comparison against
  // the plan's copy then replaces the range check, the hash and the comparison in the plan search below.
  const bool known = h->have_plan && h->from_jacobian && h->plan.n == n && (int)h->sidx.size() == m_total &&
                     (int)h->Ji.size() == jnnz && memcmp(h->Jp.data(), j_colptr, (size_t)(n + 1) * sizeof(int)) == 0 &&
                     (jnnz == 0 || memcmp(h->Ji.data(), j_rowidx, (size_t)jnnz * sizeof(int)) == 0);
  if (!known)
    for (int q = 0; q < jnnz; ++q)
      if (j_rowidx[q] < 0 || j_rowidx[q] >= m_total) {
        h->error = "hipfact_assemble_kkt: Jacobian row index out of range";
        return HIPFACT_EINVAL;
      }
  const size_t cap = (size_t)n + jnnz + nav;  // reserve_aug_jac (standard_aug_jac.c:106-133)
  hipStream_t st = h->stream;
  HCHECK(h, hipStreamSynchronize(st));  // the buffers below may still be read by queued work
  const bool want_arrays = k_colptr || k_rowidx || k_vals;
  // ---- the Jacobian and the working-set maps go to the device (the pattern only when it changed)
  const unsigned long long jhash =
      known ? h->key_hash : hash_ints(j_rowidx, (size_t)jnnz, hash_ints(j_colptr, (size_t)n + 1));
  const bool same_pattern = h->jdev_valid && h->jdev_hash == jhash && h->jdev_n == n && h->jdev_nnz == jnnz;
  HCHECK(h, h->d_jp.ensure((size_t)(n + 1) * sizeof(int)));
  HCHECK(h, h->d_ji.ensure(std::max<size_t>((size_t)jnnz * sizeof(int), 16)));
  HCHECK(h, h->d_jx.ensure(std::max<size_t>((size_t)jnnz * sizeof(double), 16)));
  HCHECK(h, h->d_vi.ensure(std::max<size_t>((size_t)n * sizeof(int), 16)));
  HCHECK(h, h->d_ci.ensure(std::max<size_t>((size_t)m_total * sizeof(int), 16)));
  if (!same_pattern) {
    HCHECK(h, hipMemcpyAsync(h->d_jp.p, j_colptr, (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    if (jnnz > 0) HCHECK(h, hipMemcpyAsync(h->d_ji.p, j_rowidx, (size_t)jnnz * sizeof(int), hipMemcpyHostToDevice, st));
    h->jdev_valid = true;
    h->jdev_hash = jhash;
    h->jdev_n = n;
    h->jdev_nnz = jnnz;
  }
  if (jnnz > 0) HCHECK(h, hipMemcpyAsync(h->d_jx.p, j_vals, (size_t)jnnz * sizeof(double), hipMemcpyHostToDevice, st));
  if (n > 0) HCHECK(h, hipMemcpyAsync(h->d_vi.p, var_index, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
  if (m_total > 0)
    HCHECK(h, hipMemcpyAsync(h->d_ci.p, cons_index, (size_t)m_total * sizeof(int), hipMemcpyHostToDevice, st));
  // ---- fill_aug_jac on the device: only when the caller asks for K itself, or on the plain path
  const bool superset = h->assemble_superset && n > 0;
  std::vector<int> kp, ki;
  int nnz = 0;
  if (want_arrays || !superset) {
    HCHECK(h, h->d_cnt.ensure(std::max<size_t>((size_t)n * sizeof(int), 16)));
    HCHECK(h, h->d_akp.ensure((size_t)(N + 1) * sizeof(int)));
    HCHECK(h, h->d_aki.ensure(std::max<size_t>(cap * sizeof(int), 16)));
    HCHECK(h, h->d_akx.ensure(std::max<size_t>(cap * sizeof(double), 16)));
    if (n > 0)
      hipLaunchKernelGGL(k_asm_count, dim3(nblocks(n)), dim3(FB), 0, st, n, h->d_jp.as<int>(), h->d_ji.as<int>(),
                         h->d_vi.as<int>(), h->d_ci.as<int>(), h->d_cnt.as<int>());
    hipLaunchKernelGGL(k_asm_scan, dim3(1), dim3(1024), 0, st, n, N, h->d_cnt.as<int>(), h->d_akp.as<int>());
    if (n > 0)
      hipLaunchKernelGGL(k_asm_fill, dim3(nblocks(n)), dim3(FB), 0, st, n, h->d_jp.as<int>(), h->d_ji.as<int>(),
                         h->d_jx.as<double>(), h->d_vi.as<int>(), h->d_ci.as<int>(), h->d_akp.as<int>(),
                         h->d_aki.as<int>(), h->d_akx.as<double>());
    HCHECK(h, hipGetLastError());
    kp.resize((size_t)N + 1);
    HCHECK(h, hipMemcpyAsync(kp.data(), h->d_akp.p, (size_t)(N + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HCHECK(h, hipStreamSynchronize(st));
    nnz = kp[N];
    if ((size_t)nnz > cap) {
      h->error = "hipfact_assemble_kkt: internal count mismatch";
      return HIPFACT_EINTERNAL;
    }
    ki.resize((size_t)nnz);
    if (nnz > 0) HCHECK(h, hipMemcpy(ki.data(), h->d_aki.p, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost));
    if (k_colptr) memcpy(k_colptr, kp.data(), (size_t)(N + 1) * sizeof(int));
    if (k_rowidx && nnz > 0) memcpy(k_rowidx, ki.data(), (size_t)nnz * sizeof(int));
    if (k_vals && nnz > 0) HCHECK(h, hipMemcpy(k_vals, h->d_akx.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
  } else if (k_nnz) {
    nnz = n + nav;
    for (int q = 0; q < jnnz; ++q) nnz += (cons_index[j_rowidx[q]] >= 0);
  }
  if (k_nnz) *k_nnz = nnz;
  if (!superset) {
    // plain path: K's own pattern is analysed (values of the unit diagonal are 1 by construction)
    if ((rc = ensure_plan(h, N, kp.data(), ki.data(), nullptr))) return rc;
    h->maps_on = false;
    h->N_ext = N;
    if (nnz > 0)
      HCHECK(h, hipMemcpyAsync(h->d_Kval.p, h->d_akx.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToDevice, st));
    return factor_and_check(h);
  }
  return superset_refactor(h, n, m_total, j_colptr, j_rowidx, cons_index, nav, nac, N, jhash, known);
}

// ---------------------------------------------------------------------------
int hipfact_reduced_matrix(hipfact_handle* h, int* nnz_out, int* colptr, int* rowidx, double* vals) {
  int rc = enter(h);
  if (rc) return rc;
  const Plan& P = h->plan;
  if (!h->have_plan || !P.saddle || h->maps_on || !nnz_out) {
    h->error = "hipfact_reduced_matrix: needs a saddle matrix set with hipfact_set_matrix under set_option(\"superset_vtable\", 0) (the exact pattern of K, no working-set maps)";
    return HIPFACT_ESTATE;
  }
  const int m = P.m;
  const long long nM = (long long)P.Mi.size();
  if (nM >= (1LL << 31)) {
    h->error = "hipfact_reduced_matrix: more than 2^31 entries";
    return HIPFACT_EINVAL;
  }
  *nnz_out = (int)nM;
  if (!colptr) return HIPFACT_OK;
  if (!rowidx || !vals) return HIPFACT_EINVAL;
  // values: the product lists once more, on the caller's (unscaled) values, into a plain array in M's order
  DevBuf d_val, d_iota;
  std::vector<double> sval((size_t)nM);
  if (nM > 0) {
    HCHECK(h, d_val.ensure((size_t)nM * sizeof(double)));
    const int nb = nblocks(nM, 1 << 16);
#define RM_LAUNCH(IDX, PK)                                                                                            \
  {                                                                                                                   \
    std::vector<IDX> iota((size_t)nM);                                                                                \
    for (long long e = 0; e < nM; ++e) iota[(size_t)e] = (IDX)e;                                                       \
    HCHECK(h, d_iota.ensure((size_t)nM * sizeof(IDX)));                                                               \
    HCHECK(h, hipMemcpyAsync(d_iota.p, iota.data(), (size_t)nM * sizeof(IDX), hipMemcpyHostToDevice, h->stream));     \
    hipLaunchKernelGGL((k_mvals_prod<IDX, PK>), dim3(nb), dim3(FB), 0, h->stream, nM, h->d_prod_ptr.as<IDX>(),        \
                       h->d_prod_a.as<int>(), h->d_prod_b.as<int>(), d_iota.as<IDX>(), h->d_Kval.as<double>(),         \
                       d_val.as<double>(), 0LL, 0, h->d_Ar_src.as<int>(), h->d_Ar_val.as<double>());                   \
    HCHECK(h, hipStreamSynchronize(h->stream));                                                                       \
  }
    if (h->idx32 && h->prod_packed)
      RM_LAUNCH(unsigned int, true)
    else if (h->idx32)
      RM_LAUNCH(unsigned int, false)
    else if (h->prod_packed)
      RM_LAUNCH(long long, true)
    else
      RM_LAUNCH(long long, false)
#undef RM_LAUNCH
    HCHECK(h, hipMemcpy(sval.data(), d_val.p, (size_t)nM * sizeof(double), hipMemcpyDeviceToHost));
  }
  // pivot order -> working-set order, lower triangle, rows ascending per column
  std::vector<int> cnt((size_t)m + 1, 0);
  for (int k = 0; k < m; ++k)
    for (long long e = P.Mp[k]; e < P.Mp[k + 1]; ++e) {
      const int r1 = P.perm[P.Mi[e]], r2 = P.perm[k];
      ++cnt[(size_t)std::min(r1, r2) + 1];
    }
  for (int j = 0; j < m; ++j) cnt[(size_t)j + 1] += cnt[j];
  std::vector<std::pair<int, double>> ent((size_t)nM);
  {
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    for (int k = 0; k < m; ++k)
      for (long long e = P.Mp[k]; e < P.Mp[k + 1]; ++e) {
        const int r1 = P.perm[P.Mi[e]], r2 = P.perm[k];
        ent[(size_t)fill[std::min(r1, r2)]++] = {std::max(r1, r2), sval[(size_t)e]};
      }
  }
  for (int j = 0; j < m; ++j) {
    std::sort(ent.begin() + cnt[j], ent.begin() + cnt[(size_t)j + 1],
              [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
    colptr[j] = cnt[j];
  }
  colptr[m] = cnt[m];
  for (long long e = 0; e < nM; ++e) {
    rowidx[e] = ent[(size_t)e].first;
    vals[e] = ent[(size_t)e].second;
  }
  return HIPFACT_OK;
}

// ---------------------------------------------------------------------------
int hipfact_spmat_create(hipfact_handle* h, int num_rows, int num_cols, const int* colptr, const int* rowidx,
                         const double* vals, hipfact_spmat** out) {
  int rc = enter(h);
  if (rc) return rc;
  if (!out || num_rows < 0 || num_cols < 0 || !colptr) return HIPFACT_EINVAL;
  *out = nullptr;
  const long long nnz = num_cols > 0 ? colptr[num_cols] : 0;
  for (int j = 0; j < num_cols; ++j)
    for (int e = colptr[j]; e < colptr[j + 1]; ++e)
      if (rowidx[e] < 0 || rowidx[e] >= num_rows) {
        h->error = "hipfact_spmat_create: row index out of range";
        return HIPFACT_EINVAL;
      }
  hipfact_spmat* M = new (std::nothrow) hipfact_spmat();
  if (!M) return HIPFACT_ENOMEM;
  M->h = h;
  h->refcount.fetch_add(1);  // released by hipfact_spmat_free (or by `fail` below)
  M->rows = num_rows;
  M->cols = num_cols;
  M->nnz = nnz;
  std::vector<int> tp(num_rows + 1, 0), ti(nnz), tsrc(nnz);
  for (long long e = 0; e < nnz; ++e) ++tp[rowidx[e] + 1];
  for (int i = 0; i < num_rows; ++i) tp[i + 1] += tp[i];
  {
    std::vector<int> fill(tp.begin(), tp.end() - 1);
    for (int j = 0; j < num_cols; ++j)
      for (int e = colptr[j]; e < colptr[j + 1]; ++e) {
        const int q = fill[rowidx[e]]++;
        ti[q] = j;
        tsrc[q] = e;
      }
  }
  auto fail = [&](int code) {
    delete M;
    h->refcount.fetch_sub(1);  // the caller still holds its own reference
    return code;
  };
  hipStream_t st = h->stream;
#define SP_UP(buf, ptr, bytes)                                                                              \
  do {                                                                                                      \
    if (M->buf.ensure(std::max<size_t>((bytes), 16)) != hipSuccess) {                                       \
      h->error = "hipfact_spmat_create: out of device memory";                                              \
      return fail(HIPFACT_ENOMEM);                                                                          \
    }                                                                                                       \
    if ((bytes) > 0 && hipMemcpyAsync(M->buf.p, (ptr), (bytes), hipMemcpyHostToDevice, st) != hipSuccess) { \
      h->error = "hipfact_spmat_create: upload failed";                                                     \
      return fail(HIPFACT_EDEVICE);                                                                         \
    }                                                                                                       \
  } while (0)
  SP_UP(cp, colptr, (size_t)(num_cols + 1) * sizeof(int));
  SP_UP(ri, rowidx, (size_t)nnz * sizeof(int));
  SP_UP(val, vals, (size_t)nnz * sizeof(double));
  SP_UP(tp, tp.data(), (size_t)(num_rows + 1) * sizeof(int));
  SP_UP(ti, ti.data(), (size_t)nnz * sizeof(int));
  SP_UP(tsrc, tsrc.data(), (size_t)nnz * sizeof(int));
#undef SP_UP
  if (M->tval.ensure(std::max<size_t>((size_t)nnz * sizeof(double), 16)) != hipSuccess ||
      M->dx.ensure(std::max<size_t>((size_t)std::max(num_rows, num_cols) * sizeof(double), 16)) != hipSuccess ||
      M->dy.ensure(std::max<size_t>((size_t)std::max(num_rows, num_cols) * sizeof(double), 16)) != hipSuccess) {
    h->error = "hipfact_spmat_create: out of device memory";
    return fail(HIPFACT_ENOMEM);
  }
  if (nnz > 0)
    hipLaunchKernelGGL(k_gather, dim3(nblocks(nnz, 1 << 16)), dim3(FB), 0, st, nnz, M->tsrc.as<int>(),
                       M->val.as<double>(), M->tval.as<double>());
  if (hipStreamSynchronize(st) != hipSuccess) {
    h->error = "hipfact_spmat_create: device error";
    return fail(HIPFACT_EDEVICE);
  }
  *out = M;
  return HIPFACT_OK;
}

int hipfact_spmat_update_values(hipfact_spmat* M, const double* vals) {
  if (!M || !vals) return HIPFACT_EINVAL;
  hipfact_handle* h = M->h;
  int rc = enter(h);
  if (rc) return rc;
  if (M->nnz == 0) return HIPFACT_OK;
  HCHECK(h, hipMemcpy(M->val.p, vals, (size_t)M->nnz * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_gather, dim3(nblocks(M->nnz, 1 << 16)), dim3(FB), 0, h->stream, M->nnz, M->tsrc.as<int>(),
                     M->val.as<double>(), M->tval.as<double>());
  HCHECK(h, hipStreamSynchronize(h->stream));
  return HIPFACT_OK;
}

int hipfact_spmat_free(hipfact_spmat** M) {
  if (M && *M) {
    hipfact_handle* owner = (*M)->h;
    (void)hipSetDevice(owner->device);
    (void)hipStreamSynchronize(owner->stream);
    delete *M;
    *M = nullptr;
    (void)hipfact_free(&owner);  // the matrix's reference to its handle
  }
  return HIPFACT_OK;
}

int hipfact_spmat_mult_device(hipfact_spmat* M, int trans, const double* d_x, double* d_y) {
  if (!M || !d_x || !d_y || trans < 0 || trans > 2) return HIPFACT_EINVAL;
  hipfact_handle* h = M->h;
  int rc = enter(h);
  if (rc) return rc;
  if (trans == 2 && M->rows != M->cols) {
    h->error = "symmetric product needs a square matrix";
    return HIPFACT_EINVAL;
  }
  const int nrows = (trans == 1) ? M->cols : M->rows;
  if (nrows == 0) return HIPFACT_OK;
  const int* ptr = (trans == 1) ? M->cp.as<int>() : M->tp.as<int>();
  const int* idx = (trans == 1) ? M->ri.as<int>() : M->ti.as<int>();
  const double* val = (trans == 1) ? M->val.as<double>() : M->tval.as<double>();
  const int* ptr2 = (trans == 2) ? M->cp.as<int>() : nullptr;
  const int* idx2 = (trans == 2) ? M->ri.as<int>() : nullptr;
  const double* val2 = (trans == 2) ? M->val.as<double>() : nullptr;
  const double avg = (double)M->nnz * (trans == 2 ? 2.0 : 1.0) / std::max(nrows, 1);
  hipStream_t st = h->stream;
  if (avg <= 2.5)
    launch_spmv<1>(st, nrows, ptr, idx, val, ptr2, idx2, val2, d_x, d_y);
  else if (avg <= 10.0)
    launch_spmv<4>(st, nrows, ptr, idx, val, ptr2, idx2, val2, d_x, d_y);
  else if (avg <= 48.0)
    launch_spmv<16>(st, nrows, ptr, idx, val, ptr2, idx2, val2, d_x, d_y);
  else
    launch_spmv<64>(st, nrows, ptr, idx, val, ptr2, idx2, val2, d_x, d_y);
  HCHECK(h, hipGetLastError());
  return HIPFACT_OK;
}

static int spmat_host_mult(hipfact_spmat* M, int trans, const double* x, double* y) {
  if (!M || !x || !y) return HIPFACT_EINVAL;
  hipfact_handle* h = M->h;
  int rc = enter(h);
  if (rc) return rc;
  const int nin = (trans == 1) ? M->rows : M->cols;
  const int nout = (trans == 1) ? M->cols : M->rows;
  if (nin > 0) HCHECK(h, hipMemcpyAsync(M->dx.p, x, (size_t)nin * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if ((rc = hipfact_spmat_mult_device(M, trans, M->dx.as<double>(), M->dy.as<double>()))) return rc;
  if (nout > 0) HCHECK(h, hipMemcpyAsync(y, M->dy.p, (size_t)nout * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  return HIPFACT_OK;
}

int hipfact_spmat_mult_vec(hipfact_spmat* M, const double* x, double* y) { return spmat_host_mult(M, 0, x, y); }
int hipfact_spmat_mult_vec_trans(hipfact_spmat* M, const double* x, double* y) { return spmat_host_mult(M, 1, x, y); }
int hipfact_spmat_mult_vec_sym(hipfact_spmat* M, const double* x, double* y) { return spmat_host_mult(M, 2, x, y); }

// ---------------------------------------------------------------------------
// three dot products in one launch + one small copy back; fixed summation order
static int cg_dots(hipfact_handle* h, int n, const double* x0, const double* y0, const double* x1, const double* y1,
                   const double* x2, const double* y2, double out[3]) {
  hipLaunchKernelGGL(k_dots3, dim3(DOT_BLOCKS), dim3(FB), 0, h->stream, n, x0, y0, x1, y1, x2, y2,
                     h->d_cg_dots.as<double>());
  HCHECK(h, hipMemcpyAsync(h->h_cg_dots.p, h->d_cg_dots.p, 3 * DOT_BLOCKS * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  const double* p = h->h_cg_dots.as<double>();
  out[0] = out[1] = out[2] = 0.0;
  for (int b = 0; b < DOT_BLOCKS; ++b)
    for (int t = 0; t < 3; ++t) out[t] += p[3 * b + t];
  return HIPFACT_OK;
}

// Hessian of the Lagrangian as an operator: an explicit matrix resident in HBM, or the caller's
// matrix-free product (SLEQP_FUNC_HESS_PROD behind sleqp_problem_hess_prod, func.c:373-408).  The
// matrix-free form moves one n-vector down and one up per product through pinned staging; every
// other vector of the Krylov loop stays on the device.
struct HessOp {
  hipfact_spmat* hess;
  hipfact_hess_prod_fn prod;
  void* user;
};

static int apply_hess(hipfact_handle* h, const HessOp& op, int n, const double* d_in, double* d_out) {
  if (op.hess) return hipfact_spmat_mult_device(op.hess, 2, d_in, d_out);
  const size_t nb = (size_t)n * sizeof(double);
  HCHECK(h, h->h_hv.ensure(2 * nb + 16));
  double* hv = h->h_hv.as<double>();
  HCHECK(h, hipMemcpyAsync(hv, d_in, nb, hipMemcpyDeviceToHost, h->stream));
  HCHECK(h, hipStreamSynchronize(h->stream));
  if (op.prod(op.user, hv, hv + n) != 0) {
    h->error = "Hessian product callback failed";
    return HIPFACT_EINTERNAL;
  }
  HCHECK(h, hipMemcpyAsync(d_out, hv + n, nb, hipMemcpyHostToDevice, h->stream));
  return HIPFACT_OK;
}

static int tr_args_ok(hipfact_handle* h, const HessOp& op, const double* gradient, double* newton_step,
                      double trust_radius, const char* who) {
  const Plan& P = h->plan;
  const int n = P.saddle ? P.n : 0;
  const bool hess_ok = op.hess ? (op.hess->h == h && op.hess->rows == n && op.hess->cols == n) : op.prod != nullptr;
  if (!P.saddle || !hess_ok || !gradient || !newton_step || !(trust_radius > 0.0)) {
    h->error = std::string(who) + ": needs a factorised saddle matrix and an n x n Hessian (explicit, on the same "
                                  "handle, or a product callback)";
    return HIPFACT_EINVAL;
  }
  return HIPFACT_OK;
}

#include "krylov_device.inc"

static int steihaug_impl(hipfact_handle* h, const HessOp& op, const double* gradient, double trust_radius,
                         double rel_tol, int max_iter, double* newton_step, double* tr_dual, int* iterations) {
  int rc;
  if ((rc = require_factor(h, "hipfact_steihaug_solve"))) return rc;
  if ((rc = tr_args_ok(h, op, gradient, newton_step, trust_radius, "hipfact_steihaug_solve"))) return rc;
  const Plan& P = h->plan;
  const int n = P.n;
  const int N = h->N_ext;
  hipStream_t st = h->stream;
  const size_t nb = (size_t)n * sizeof(double);
  HCHECK(h, h->d_cg_b.ensure((size_t)N * sizeof(double)));
  HCHECK(h, h->d_cg_z.ensure((size_t)N * sizeof(double)));
  HCHECK(h, h->d_cg_vec.ensure(4 * nb + 64));
  HCHECK(h, h->d_cg_dots.ensure(3 * DOT_BLOCKS * sizeof(double)));
  HCHECK(h, h->h_cg_dots.ensure(3 * DOT_BLOCKS * sizeof(double)));
  HCHECK(h, h->h_stage.ensure(nb));
  // r lives in the head of the KKT right-hand side [r; 0]; g is the head of the KKT solution
  double* r = h->d_cg_b.as<double>();
  const double* g = h->d_cg_z.as<double>();
  double* z = h->d_cg_vec.as<double>();
  double* d = z + n;
  double* Bd = d + n;
  double* grad = Bd + n;
  HCHECK(h, hipStreamSynchronize(st));
  memcpy(h->h_stage.p, gradient, nb);
  HCHECK(h, hipMemsetAsync(h->d_cg_b.p, 0, (size_t)N * sizeof(double), st));
  HCHECK(h, hipMemcpyAsync(r, h->h_stage.p, nb, hipMemcpyHostToDevice, st));
  HCHECK(h, hipMemcpyAsync(grad, r, nb, hipMemcpyDeviceToDevice, st));
  HCHECK(h, hipMemsetAsync(z, 0, nb, st));
  const int vb = nblocks(n);
  const double rel_tol_sq = rel_tol * rel_tol;
  const double rad_sq = trust_radius * trust_radius;
  double dots[3];
  if (tr_dual) *tr_dual = -1.0;  // SLEQP_NONE
  int it = 0;
  bool boundary = false;

  // The refinement of a projection is controlled on the device; its control block arrives with
  // the dot products that follow it.  Only when a projection needed more passes than its graph
  // carries (ill-conditioned working set) is it continued from here and the products redone.
  bool cont = false;
  // g0 = P[r0], d0 = -g0
  if ((rc = solve_async(h, h->d_cg_b.as<double>(), h->d_cg_z.as<double>()))) return rc;
  do {
    HCHECK(h, hipMemsetAsync(d, 0, nb, st));
    hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, -1.0, g, 0.0, d);
    if ((rc = cg_dots(h, n, d, d, r, g, nullptr, nullptr, dots))) return rc;
    if ((rc = finish_solve(h, &cont))) return rc;
  } while (cont);
  double d_nrm_sq = dots[0], r_dot_g = dots[1];
  double z_nrm_sq = 0.0;
  bool dev_used = false;
  if (!(d_nrm_sq < rel_tol_sq)) {
    bool touched = false;
    int dstate = 0, dits = 0;
    if ((rc = steihaug_device_loop(h, op.hess, n, r_dot_g, rel_tol_sq, rad_sq, max_iter, r, h->d_cg_z.as<double>(), z, d, Bd, grad,
                                   &dev_used, &touched, &dstate, &dits)))
      return rc;
    if (dev_used) {
      h->cg_device_runs++;
      it = dits;
      boundary = (dstate == 2);
      if (dstate == 4) HCHECK(h, hipMemsetAsync(z, 0, nb, st));  // iteration cap: the step stays cleared (see below)
    } else if (touched) {
      // a projection did not meet the tolerance unchecked: once more from the start with the host in the loop
      h->cg_device_fallbacks++;
      h->cg_device_loop = false;
      rc = steihaug_impl(h, op, gradient, trust_radius, rel_tol, max_iter, newton_step, tr_dual, iterations);
      h->cg_device_loop = true;
      return rc;
    }
  }
  if (!dev_used && !(d_nrm_sq < rel_tol_sq)) {
    for (it = 0;; ++it) {
      if (max_iter != -1 && it >= max_iter) {
        // the reference leaves newton_step cleared when the iteration cap is hit before any of
        // its exit tests (steihaug_solver.c:254,282-285): mirrored, the step is zero
        HCHECK(h, hipMemsetAsync(z, 0, nb, st));
        break;
      }
      if (fabs(r_dot_g) < rel_tol_sq) break;  // interior solution p = z
      // B d, d^T B d, z^T d
      if ((rc = apply_hess(h, op, n, d, Bd))) return rc;
      if ((rc = cg_dots(h, n, d, Bd, z, d, d, d, dots))) return rc;
      const double dBd = dots[0], z_dot_d = dots[1];
      d_nrm_sq = dots[2];
      if (dBd <= 0.0) {
        // negative curvature: go to the boundary along d, pick the better of the two intersections
        double e2[3];
        if ((rc = cg_dots(h, n, grad, d, z, Bd, nullptr, nullptr, e2))) return rc;
        const double gd = e2[0], zBd = e2[1];
        const double inner = z_dot_d * z_dot_d - d_nrm_sq * (z_nrm_sq - rad_sq);
        const double tau_min = 1. / d_nrm_sq * (-z_dot_d - sqrt(inner));
        const double tau_max = 1. / d_nrm_sq * (-z_dot_d + sqrt(inner));
        const double obj_min = tau_min * ((gd + zBd) + 0.5 * tau_min * dBd);
        const double obj_max = tau_max * ((gd + zBd) + 0.5 * tau_max * dBd);
        const double tau = (obj_min < obj_max) ? tau_min : tau_max;
        hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, tau, d, 1.0, z);
        break;
      }
      const double alpha = r_dot_g / dBd;
      const double z_next_nrm_sq = z_nrm_sq + 2.0 * alpha * z_dot_d + alpha * alpha * d_nrm_sq;
      if (z_next_nrm_sq >= rad_sq) {
        // sleqp_tr_compute_bdry_sol (tr/tr_util.c:8-58)
        const double inner = z_dot_d * z_dot_d - d_nrm_sq * (z_nrm_sq - rad_sq);
        const double factor = 1. / d_nrm_sq * (-z_dot_d + sqrt(inner));
        hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, factor, d, 1.0, z);
        boundary = true;
        break;
      }
      hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, alpha, d, 1.0, z);   // z += alpha d
      hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, alpha, Bd, 1.0, r);  // r += alpha B d
      z_nrm_sq = z_next_nrm_sq;
      if ((rc = solve_async(h, h->d_cg_b.as<double>(), h->d_cg_z.as<double>()))) return rc;  // g = P[r]
      do {
        if ((rc = cg_dots(h, n, r, g, nullptr, nullptr, nullptr, nullptr, dots))) return rc;
        if ((rc = finish_solve(h, &cont))) return rc;
      } while (cont);
      const double beta = dots[0] / r_dot_g;
      r_dot_g = dots[0];
      hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, -1.0, g, beta, d);  // d = -g + beta d
    }
  }
  if (boundary && tr_dual) {
    // steihaug_tr_dual (steihaug_solver.c:187-216)
    if ((rc = apply_hess(h, op, n, z, Bd))) return rc;
    if ((rc = cg_dots(h, n, z, Bd, z, grad, nullptr, nullptr, dots))) return rc;
    const double comb = dots[0] + dots[1];
    *tr_dual = comb < 0.0 ? (-comb) / rad_sq : 0.0;
  }
  HCHECK(h, hipGetLastError());
  HCHECK(h, hipMemcpyAsync(h->h_stage.p, z, nb, hipMemcpyDeviceToHost, st));
  if ((rc = check_info(h, "solve"))) return rc;  // synchronises; a timed-out sweep invalidates the step
  memcpy(newton_step, h->h_stage.p, nb);
  if (iterations) *iterations = it;
  return HIPFACT_OK;
}

int hipfact_steihaug_solve(hipfact_handle* h, hipfact_spmat* hess, const double* gradient, double trust_radius,
                           double rel_tol, int max_iter, double* newton_step, double* tr_dual, int* iterations) {
  int rc = enter(h);
  if (rc) return rc;
  if (!hess) {
    h->error = "hipfact_steihaug_solve: no Hessian";
    return HIPFACT_EINVAL;
  }
  const HessOp op = {hess, nullptr, nullptr};
  return steihaug_impl(h, op, gradient, trust_radius, rel_tol, max_iter, newton_step, tr_dual, iterations);
}

// Generalised Lanczos trust-region method (GLTR: Gould, Lucidi, Roma, Toint 1999), the algorithm the
// reference runs through trlib_krylov_min (tr/trlib_solver.c:322-352) with the null-space projection as
// preconditioner.  Preconditioned Lanczos with M^-1 = P (the factorised KKT projection of this handle):
//     t_0 = g,  y_k = P t_k,  gamma_k = sqrt(t_k . y_k),  q_k = y_k / gamma_k
//     delta_k = q_k . H q_k
//     t_{k+1} = H q_k - (delta_k / gamma_k) t_k - (gamma_k / gamma_{k-1}) t_{k-1}
// and after every step the trust-region problem on the tridiagonal T_k (host, tridiag_tr.cpp):
//     min 1/2 h^T T_k h + gamma_0 e_1^T h,  ||h|| <= radius,        s = Q_k h.
// While the solution is interior this is the CG iterate; on the boundary the iteration simply goes on
// (Steihaug stops there) until gamma_{k+1} |h_k| <= max(abs_tol, rel_tol gamma_0) - trlib's tests for
// the interior and the boundary case (tol_rel_i / tol_rel_b, trlib_solver.c:272-275).  Negative curvature
// and the hard case are handled inside the tridiagonal solve.  All n-vectors (t, y, H q and the basis Q,
// n x (iterations + 1)) stay in HBM; the host sees two scalars per iteration.
static int gltr_impl(hipfact_handle* h, const HessOp& op, const double* gradient, double trust_radius, double rel_tol,
                     int max_iter, double* newton_step, double* tr_dual, int* iterations) {
  int rc;
  if ((rc = require_factor(h, "hipfact_tr_solve"))) return rc;
  if ((rc = tr_args_ok(h, op, gradient, newton_step, trust_radius, "hipfact_tr_solve"))) return rc;
  const Plan& P = h->plan;
  const int n = P.n;
  const int N = h->N_ext;
  hipStream_t st = h->stream;
  const size_t nb = (size_t)n * sizeof(double);
  const int cap = (max_iter >= 0 ? std::min(max_iter, n) : std::min(n, 1000)) + 1;
  HCHECK(h, h->d_lz_Q.ensure(std::max<size_t>((size_t)cap * nb, 16)));
  HCHECK(h, h->d_lz_b.ensure(std::max<size_t>((size_t)3 * N * sizeof(double), 16)));
  HCHECK(h, h->d_lz_coef.ensure((size_t)cap * sizeof(double)));
  HCHECK(h, h->d_cg_z.ensure(std::max<size_t>((size_t)N * sizeof(double), 16)));
  HCHECK(h, h->d_cg_vec.ensure(4 * nb + 64));
  HCHECK(h, h->d_cg_dots.ensure(3 * DOT_BLOCKS * sizeof(double)));
  HCHECK(h, h->h_cg_dots.ensure(3 * DOT_BLOCKS * sizeof(double)));
  HCHECK(h, hipStreamSynchronize(st));
  HCHECK(h, h->h_stage.ensure(std::max(nb, (size_t)cap * sizeof(double))));
  double* Q = h->d_lz_Q.as<double>();
  double* tb[3] = {h->d_lz_b.as<double>(), h->d_lz_b.as<double>() + N, h->d_lz_b.as<double>() + 2 * (size_t)N};
  const double* y = h->d_cg_z.as<double>();  // head of the KKT solution = P t
  double* Hq = h->d_cg_vec.as<double>();
  double* s = Hq + n;
  const int vb = nblocks(n);
  if (tr_dual) *tr_dual = 0.0;
  if (iterations) *iterations = 0;
  if (n == 0) return HIPFACT_OK;
  memcpy(h->h_stage.p, gradient, nb);
  HCHECK(h, hipMemsetAsync(h->d_lz_b.p, 0, (size_t)3 * N * sizeof(double), st));  // the tails [.; 0] stay zero
  HCHECK(h, hipMemcpyAsync(tb[0], h->h_stage.p, nb, hipMemcpyHostToDevice, st));
  std::vector<double> delta, gamma, hvec;
  double dots[3];
  bool cont = false;
  // y_0 = P t_0, gamma_0
  if ((rc = solve_async(h, tb[0], h->d_cg_z.as<double>()))) return rc;
  do {
    if ((rc = cg_dots(h, n, tb[0], y, nullptr, nullptr, nullptr, nullptr, dots))) return rc;
    if ((rc = finish_solve(h, &cont))) return rc;
  } while (cont);
  const double gamma0 = dots[0] > 0.0 ? sqrt(dots[0]) : 0.0;
  auto finish_zero = [&]() {
    memset(newton_step, 0, nb);
    return HIPFACT_OK;
  };
  if (!(gamma0 > 0.0) || !(gamma0 < 1.7e308)) {
    if (gamma0 == 0.0) return finish_zero();
    h->error = "hipfact_tr_solve: non-finite projected gradient";
    return HIPFACT_EINTERNAL;
  }
  const double tol = rel_tol * gamma0;
  hipLaunchKernelGGL(k_scale_to, dim3(vb), dim3(FB), 0, st, n, 1.0 / gamma0, y, Q);
  // Only P t_k ever enters the recurrence, so t_k is replaced by y_k = P t_k once it has been projected
  // (the "residual update" of Gould, Hribar, Nocedal 2001): the components of H q in the range of A^T would
  // otherwise pile up in t, and the projection of a vector that is mostly range space loses the accuracy
  // of its null-space part.
  HCHECK(h, hipMemcpyAsync(tb[0], y, nb, hipMemcpyDeviceToDevice, st));
  gamma.push_back(gamma0);  // gamma[k] = ||t_k||_P; gamma[0] is not part of T
  double lambda = 0.0;
  int cur = 0, k = 0;  // tb[cur] = t_k, tb[(cur + 2) % 3] = t_{k-1}
  const int kmax = cap - 1;
  for (k = 0; k < kmax; ++k) {
    const double* q = Q + (size_t)k * n;
    if ((rc = apply_hess(h, op, n, q, Hq))) return rc;
    if ((rc = cg_dots(h, n, q, Hq, nullptr, nullptr, nullptr, nullptr, dots))) return rc;
    delta.push_back(dots[0]);
    // T_k is complete: trust-region problem on the tridiagonal
    hvec.assign((size_t)k + 1, 0.0);
    if (tridiag_tr_solve(k + 1, delta.data(), gamma.data(), gamma0, trust_radius, hvec.data(), &lambda) != 0) {
      h->error = "hipfact_tr_solve: tridiagonal trust-region subproblem failed";
      return HIPFACT_EINTERNAL;
    }
    // t_{k+1} = H q_k - (delta_k / gamma_k) t_k - (gamma_k / gamma_{k-1}) t_{k-1}
    double* tn = tb[(cur + 1) % 3];
    HCHECK(h, hipMemcpyAsync(tn, Hq, nb, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, -delta[k] / gamma[k], tb[cur], 1.0, tn);
    if (k > 0)
      hipLaunchKernelGGL(k_axpby, dim3(vb), dim3(FB), 0, st, n, -gamma[k] / gamma[k - 1], tb[(cur + 2) % 3], 1.0, tn);
    if ((rc = solve_async(h, tn, h->d_cg_z.as<double>()))) return rc;
    do {
      if ((rc = cg_dots(h, n, tn, y, nullptr, nullptr, nullptr, nullptr, dots))) return rc;
      if ((rc = finish_solve(h, &cont))) return rc;
    } while (cont);
    const double gnext = dots[0] > 0.0 ? sqrt(dots[0]) : 0.0;
    if (!(gnext == gnext)) {
      h->error = "hipfact_tr_solve: non-finite Lanczos vector";
      return HIPFACT_EINTERNAL;
    }
    gamma.push_back(gnext);
    HCHECK(h, hipMemcpyAsync(tn, y, nb, hipMemcpyDeviceToDevice, st));  // t_{k+1} := P t_{k+1}
    cur = (cur + 1) % 3;
    // converged (interior: CG residual; boundary: trlib's test), or the Krylov space is exhausted
    if (gnext * fabs(hvec[k]) <= tol || gnext <= 1e-14 * gamma0) {
      ++k;
      break;
    }
    if (k + 1 < kmax) hipLaunchKernelGGL(k_scale_to, dim3(vb), dim3(FB), 0, st, n, 1.0 / gnext, y, Q + (size_t)(k + 1) * n);
  }
  const int dim = (int)hvec.size();
  if (dim == 0) return finish_zero();
  // s = Q h
  HCHECK(h, hipStreamSynchronize(st));
  memcpy(h->h_stage.p, hvec.data(), (size_t)dim * sizeof(double));
  HCHECK(h, hipMemcpyAsync(h->d_lz_coef.p, h->h_stage.p, (size_t)dim * sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_combine, dim3(vb), dim3(FB), 0, st, n, dim, Q, h->d_lz_coef.as<double>(), s);
  HCHECK(h, hipGetLastError());
  HCHECK(h, hipStreamSynchronize(st));
  HCHECK(h, hipMemcpyAsync(h->h_stage.p, s, nb, hipMemcpyDeviceToHost, st));
  if ((rc = check_info(h, "solve"))) return rc;  // synchronises
  memcpy(newton_step, h->h_stage.p, nb);
  if (tr_dual) *tr_dual = lambda;
  if (iterations) *iterations = std::min(k, dim);
  return HIPFACT_OK;
}

int hipfact_tr_solve(hipfact_handle* h, int method, hipfact_spmat* hess, hipfact_hess_prod_fn prod, void* user,
                     const double* gradient, double trust_radius, double rel_tol, int max_iter, double* newton_step,
                     double* tr_dual, int* iterations) {
  int rc = enter(h);
  if (rc) return rc;
  if (!hess && !prod) {
    h->error = "hipfact_tr_solve: neither an explicit Hessian nor a product callback";
    return HIPFACT_EINVAL;
  }
  const HessOp op = {hess, hess ? nullptr : prod, user};
  if (method == HIPFACT_TR_STEIHAUG)
    return steihaug_impl(h, op, gradient, trust_radius, rel_tol, max_iter, newton_step, tr_dual, iterations);
  if (method == HIPFACT_TR_GLTR)
    return gltr_impl(h, op, gradient, trust_radius, rel_tol, max_iter, newton_step, tr_dual, iterations);
  h->error = "hipfact_tr_solve: unknown method";
  return HIPFACT_EINVAL;
}

// ---------------------------------------------------------------------------
int hipfact_set_option(hipfact_handle* h, const char* name, double value) {
  if (!h || !name) return HIPFACT_EINVAL;
  if (!strcmp(name, "refine_steps")) {
    h->refine_steps = std::max(0, (int)value);
    h->refine_inline = h->refine_steps;
    drop_graphs(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "refine_max")) {
    h->refine_max = std::max(0, (int)value);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "refine_adaptive")) {
    h->refine_adaptive = value != 0.0;
    drop_graphs(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "refine_tol")) {
    h->refine_tol = value;
    drop_graphs(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "fail_omega")) {
    h->fail_omega = value;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "equilibrate")) {  // takes effect at the next factorisation
    h->equilibrate = value != 0.0;
    drop_graphs(h);
    h->factored = false;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "use_graph")) {
    h->use_graph = value != 0.0;
    if (!h->use_graph) drop_graphs(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "top_max_fronts")) {  // 0 disables the single-launch top-of-tree solve
    h->top_max_fronts = (int)value;
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "factor_top_max")) {  // 0: one launch per phase and level everywhere
    h->factor_top_max = (int)value;
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "wide_min_rows")) {
    h->wide_min_rows = (int)value;
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "top_prefetch")) {
    h->top_prefetch = value != 0.0;
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "pull_max_children")) {  // 0: extend-add always through the separate assembly kernel
    h->pull_max_children = (int)value;
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "split_max_fronts")) {
    h->split_max_fronts = (int)value;
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "debug_fake_timeout")) {  // test hook for the fallback to the per-level launches
    h->fake_timeouts = (int)value;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "factor_top_levels")) {
    h->factor_top_levels = std::max(0, (int)value);
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "solve_slices")) {  // 0: one item per front in the fused solve launch (fronts of up to 1024 rows only)
    if (h->solve_slices != (value != 0.0)) invalidate_plans(h);
    h->solve_slices = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "speculate")) {
    h->speculate = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "chain_fuse")) {
    if (h->chain_fuse != (value != 0.0)) invalidate_plans(h);
    h->chain_fuse = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "cg_device_loop")) {  // 0: the host reads the dot products of every CG iteration (steihaug_impl)
    h->cg_device_loop = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "xupd_fused")) {  // 0: x = b_x - A^T y as a launch of its own behind the tree (k_x_saddle)
    if (h->xupd_fused != (value != 0.0)) drop_graphs(h);
    h->xupd_fused = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "refine_check_every")) {  // residual check on every k-th solve of a well-conditioned factorisation
    h->refine_check_every = std::max(1, (int)value);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "decide_lazy")) {  // 0: every solve graph ends with its own verdict launch
    if (h->decide_lazy != (value != 0.0)) {
      flush_decide(h);
      drop_graphs(h);
    }
    h->decide_lazy = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "rhs_fused")) {  // 0: k_rhs_saddle in front of the single-launch solve
    if (h->rhs_fused != (value != 0.0)) drop_graphs(h);
    h->rhs_fused = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "spanel_fold")) {  // 0: the solve panels in a launch of their own behind the factorisation
    if (h->spanel_fold != (value != 0.0)) invalidate_plans(h);
    h->spanel_fold = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "spanel_fold_room")) {
    if (h->spanel_fold_room != (int)value) invalidate_plans(h);
    h->spanel_fold_room = (int)value;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "spanel_side")) {
    h->spanel_side = value != 0.0;
    drop_graphs(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "solve_fused")) {  // 0: the two-launch / per-level solve kernels on the factor panels
    h->solve_fused = value != 0.0;
    invalidate_plans(h);
    return HIPFACT_OK;
  }
  if (!strcmp(name, "zero_behind")) {  // solve-panel items zero the bottom levels' panels behind them (plan option)
    if (h->zero_behind != (value != 0.0)) invalidate_plans(h);
    h->zero_behind = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "superset_vtable")) {  // 0: hipfact_set_matrix analyses the pattern of K itself (exact-pattern plan cache only)
    h->superset_vtable = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "assemble_superset")) {
    h->assemble_superset = value != 0.0;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "plan_cache")) {  // inactive plan states kept (LRU); 0: one pattern at a time
    h->plan_cache_max = std::max(0, (int)value);
    while ((int)h->cache.size() > h->plan_cache_max) h->cache.pop_back();
    return HIPFACT_OK;
  }
  if (!strcmp(name, "debug_phases")) {
    h->debug_phases = (int)value;
    return HIPFACT_OK;
  }
  if (!strcmp(name, "profile")) {  // event-time every kernel class; value < 0 resets the counters
    if (h->prof.on) prof_collect(h);
    if (value < 0)
      for (int c = 0; c < PC_COUNT; ++c) h->prof.ms[c] = 0, h->prof.cnt[c] = 0;
    h->prof.on = value > 0;
    return HIPFACT_OK;
  }
  bool plan_opt = true;
  if (!strcmp(name, "ordering"))
    h->prm.ordering = (int)value;
  else if (!strcmp(name, "wmax"))
    h->prm.wmax = (int)value;
  else if (!strcmp(name, "max_children"))
    h->prm.max_children = (int)value;
  else if (!strcmp(name, "nd_leaf"))
    h->prm.nd_leaf = (int)value;
  else if (!strcmp(name, "nd_sep_frac"))
    h->prm.nd_sep_frac = value;
  else if (!strcmp(name, "force_generic"))
    h->prm.force_generic = value != 0.0;
  else if (!strcmp(name, "dense_tau"))  // a Jacobian column is dense from max(dense_min, dense_tau sqrt(m)) entries
    h->prm.dense_tau = value;
  else if (!strcmp(name, "dense_min"))
    h->prm.dense_min = std::max(1, (int)value);
  else if (!strcmp(name, "dense_max"))  // at most this many (<= 64; 0: no dense-column treatment)
    h->prm.dense_max = std::min(64, std::max(0, (int)value));
  else if (!strcmp(name, "adopt_leaves"))
    h->prm.adopt_leaves = value != 0.0;
  else
    plan_opt = false;
  if (plan_opt) {
    invalidate_plans(h);  // next set_matrix re-analyses
    return HIPFACT_OK;
  }
  h->error = std::string("unknown option: ") + name;
  return HIPFACT_EINVAL;
}

int hipfact_debug_copy(hipfact_handle* h, const char* name, void* out, size_t bytes) {
  int rc = enter(h);
  if (rc) return rc;
  if (!name || !out) return HIPFACT_EINVAL;
  const DevBuf* b = nullptr;
  if (!strcmp(name, "L")) b = &h->d_L;
  else if (!strcmp(name, "SPf")) b = &h->d_SPf;
  else if (!strcmp(name, "SPb")) b = &h->d_SPb;
  else if (!strcmp(name, "sitems")) b = &h->d_sitems;
  else if (!strcmp(name, "y")) b = &h->d_y;
  else if (!strcmp(name, "xhat")) b = &h->d_xhat;
  else if (!strcmp(name, "ysol")) b = &h->d_ysol;
  else if (!strcmp(name, "uvec")) b = &h->d_uvec;
  else if (!strcmp(name, "dscale")) b = &h->d_dscale;
  if (!b || !b->p || bytes > b->bytes) {
    h->error = "hipfact_debug_copy: unknown buffer or size";
    return HIPFACT_EINVAL;
  }
  HCHECK(h, hipStreamSynchronize(h->stream));
  HCHECK(h, hipMemcpy(out, b->p, bytes, hipMemcpyDeviceToHost));
  return HIPFACT_OK;
}

int hipfact_get_info(const hipfact_handle* h, const char* name, double* value) {
  if (!h || !name || !value) return HIPFACT_EINVAL;
  const Plan& P = h->plan;
  if (!strncmp(name, "prof_", 5)) {  // prof_<class>_ms / prof_<class>_count
    prof_collect(const_cast<hipfact_handle*>(h));
    for (int c = 0; c < PC_COUNT; ++c) {
      const size_t len = strlen(kProfNames[c]);
      if (!strncmp(name + 5, kProfNames[c], len) && name[5 + len] == '_') {
        if (!strcmp(name + 6 + len, "ms")) {
          *value = h->prof.ms[c];
          return HIPFACT_OK;
        }
        if (!strcmp(name + 6 + len, "count")) {
          *value = (double)h->prof.cnt[c];
          return HIPFACT_OK;
        }
      }
    }
    return HIPFACT_EINVAL;
  }
#define INFO(key, expr)       \
  if (!strcmp(name, key)) {   \
    *value = (double)(expr);  \
    return HIPFACT_OK;        \
  }
  INFO("N", h->have_plan ? h->N_ext : P.N) INFO("n", P.n) INFO("m", P.m) INFO("saddle", P.saddle) INFO("nnzK", P.nnzK) INFO("nnzL", P.nnzL)
  INFO("nnzL_true", P.nnzL_true) INFO("flops", P.flops) INFO("flops_dense", P.flops_dense) INFO("nsuper", P.nsuper)
  INFO("nlevels", P.nlevels) INFO("nprod", P.nprod) INFO("L_bytes", P.L_size * 8.0) INFO("U_bytes", P.U_size * 8.0)
  INFO("analysis_s", P.t_total) INFO("order_s", P.t_order) INFO("symbolic_s", P.t_symbolic)
  INFO("num_zero_pivots", h->info_host[INFO_ZERO_PIVOT]) INFO("num_neg_pivots", h->info_host[INFO_NEG_PIVOT])
  INFO("cache_hits", h->cache_hits) INFO("plan_swaps", h->plan_swaps) INFO("plans_cached", h->cache.size())
  INFO("no_dataflow", h->no_dataflow) INFO("dataflow_fallbacks", h->dataflow_fallbacks) INFO("fused_solve", h->fused_solve) INFO("spanel_folded", h->sp_folded) INFO("solve_items", h->n_sitems) INFO("chain_levels_fused", [&] { int c = 0; for (const LevelInfo& li : h->levels) c += li.mini_cnt > 0; return c; }()) INFO("solve_panel_bytes", h->sp_bytes) INFO("N_internal", P.N) INFO("maps_on", h->maps_on) INFO("m_struct", h->m_struct) INFO("analyses", h->analyses) INFO("num_factor", h->num_factor) INFO("cg_device_runs", h->cg_device_runs) INFO("cg_device_fallbacks", h->cg_device_fallbacks) INFO("dense_columns", h->nd) INFO("vtable_rows", h->vj->rows()) INFO("vtable_retries", h->vtable_retries) INFO("superset_vtable", h->superset_vtable)
  INFO("num_solve", h->num_solve) INFO("num_refined", h->num_refined) INFO("refine_adaptive", h->refine_adaptive)
  INFO("num_passes", h->num_passes) INFO("last_omega", h->last_ctl.omega) INFO("last_iters", h->last_ctl.iters)
  INFO("last_status", h->last_ctl.status) INFO("last_tol", h->last_ctl.tol) INFO("kappa_est", h->last_ctl.kappa)
  INFO("refine_inline", h->refine_inline) INFO("refine_tol", h->refine_tol) INFO("equilibrate", h->equilibrate)
  INFO("factor_top_level", h->ftop_level) INFO("factor_top_count", h->ftop_count) INFO("top_level", h->top_level) INFO("top_count", h->top_count) INFO("solve_timeouts", h->h_info.p ? h->h_info.as<int>()[INFO_TIMEOUT] : 0)
  INFO("use_graph", h->use_graph) INFO("num_graphs", h->graphs.size()) INFO("max_r", P.max_r) INFO("max_w", P.max_w) INFO("refine_steps", h->refine_steps)
  INFO("device", h->device) INFO("nnzM", P.Mi.size()) INFO("nnzA", P.Ar_src.size())
  INFO("rows_total", P.sn_rows.size()) INFO("ent_fused", h->ent_fused) INFO("ent_split", h->ent_split)
  INFO("rows_fused", h->rows_fused) INFO("rows_split", h->rows_split)
#undef INFO
  return HIPFACT_EINVAL;
}

}  // extern "C"

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
