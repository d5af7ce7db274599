// Plain-old-data descriptors shared by the host runtime and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace hipfact {

// One front (supernode) of the multifrontal LDL^T.
//   panel  : L arena + Loff, r x w column-major (ld = r).  After factorisation the
//            top w x w block holds inv(L11) strictly below the diagonal and the
//            pivots d on the diagonal; rows w..r hold L21.
//   update : U arena + Uoff, u x u column-major (ld = u), lower triangle valid.
//   uvec   : solve workspace + uoff, u doubles (update vector passed to the parent).
struct SnDesc {
  long long Loff;
  long long Uoff;
  long long uoff;
  long long rowoff;  // into rows[]: r sorted pivot-order row indices, own columns first
  long long reloff;  // into rel[]: u positions inside the parent's front
  int c0;            // first column (pivot order)
  int w;             // columns
  int r;             // rows of the front
  int parent;        // -1 for a root
  int child_begin;   // children in child_idx[child_begin, child_end)
  int child_end;
  int pad0;  // elimination-tree level of the front
  int pad1;  // offset of this front's inverse relative indices (as a child) in inv[]
};

// children of a front, flattened for the pull-mode extend-add (one uniform load per workgroup)
constexpr int MAXCH = 4;
struct PullDesc {
  long long Uoff[MAXCH];    // child's update matrix
  long long reloff[MAXCH];  // child's relative indices
  int invoff[MAXCH];        // child's inverse relative indices (SnDesc::pad1)
  int uc[MAXCH];            // order of the child's update matrix
  int n;                    // number of children in this block (<= MAXCH)
  int next;                 // index of the block with the front's next children in the overflow array, -1: none
  int pad[2];
};

// one workgroup of a split-phase kernel: everything it needs in ONE uniform load (a dependent
// walk items -> descriptor -> children costs a memory round trip per hop on the critical path)
struct FrontItem {
  long long Loff, Uoff;
  int w, r;
  int part;    // panel row block / (I << 16 | J) tile / unused
  int nchild;  // children of the front (assign-mode Schur update when 0)
  PullDesc pd;
};

// k_front_schur_pair: the second front b of a pair of chain fronts forms its update matrix from the grandchild's
// (U_g, leading dimension u_g, entry (i, j) of U_b at (i + off_g, j + off_g)) with the panels of a (its child) and b
struct PairDesc {
  long long Loff_a, Uoff_g;
  int w_a, r_a, rowoff_a;  // rows of L21_a that belong to U_b's rows start at rowoff_a (= w_b)
  int u_g, off_g;
};

// one workgroup of the single-launch top-of-tree factorisation kernel
struct TopFItem {
  FrontItem it;
  int role;   // 0 pivot, 1 panel, 2 Schur, 3 solve panels, 4 panel rows + Schur tile in one (kernels_front_fused.inc),
              // 5 whole front (experiment), 6 pivot block + panel rows in one
  int front;  // index of the front's counters
  int part2;  // Schur: the second team's tile
  int nwait;  // number of children
  int wait_id[MAXCH];   // children
  int wait_cnt[MAXCH];  // Schur workgroups of each child in this launch (0: finished before the launch)
  int target;           // Schur: panel workgroups of the own front
  int crows;            // Schur: 64 = one 32 x 32 tile per workgroup (narrow levels), 128 = two teams with a 64 x 64 tile each
  int prows;            // panel: rows per workgroup (128, or 64 with two waves per 16-row strip)
  int sidx, scount;     // Schur: index among / number of the Schur workgroups of the front
  int post;             // the pivot workgroup posts inv(L11) tile by tile, the panel workgroups poll for it (no flag hop)
  long long xoff;       // the front's wp x wp slot in the arena of posted pivot blocks
  // The parent's PIVOT block only reads the leading k x k corner of a child's update matrix (k = the child's update
  // rows that are pivot columns of the parent): the Schur workgroups whose tiles touch that corner ("head") come first
  // in the level's order and count themselves in a second counter (hdone), and the parent's pivot workgroup waits for
  // that one - its chain runs beside the child's remaining tiles.  Panel / Schur / fused roles of the parent still wait
  // for the whole child (they read the rest, and the update arena is reused on that assumption).
  int head;               // Schur: this workgroup's tile(s) touch the head corner
  int wait_head[MAXCH];   // pivot: head workgroups of each child (0: wait for the whole child, wait_cnt)
};

// one front of the single-launch top-of-tree solve kernels (one uniform load per workgroup)
struct TopItem {
  long long Loff, uoff, rowoff;
  int s, c0, w, r, parent;
  int nchild;              // -1: more than MAXCH children (generic path through the SnDesc walk)
  int prefetch;            // bit 0 / 1: the forward / backward step prefetches (buffers fit the LDS)
  int pad;
  long long c_uoff[MAXCH];   // children's update vectors
  long long c_reloff[MAXCH]; // children's relative indices
  int c_uc[MAXCH];
  int c_id[MAXCH];
  int c_wait[MAXCH];         // child is part of the same launch: wait until its flag reaches this count (0: no wait)
  // wide fronts (thousands of update rows): one "head" workgroup (pivot block) and several "slice"
  // workgroups (256 update rows each) instead of one workgroup that streams the whole panel
  int kind;                  // 0 ordinary front, 1 head, 2 slice
  int a0, a1;                // slice: update rows [a0, a1)
  int nsl;                   // number of slices of the front
  long long poff;            // backward: the front's partial sums (nsl x w) in the scratch buffer
  int c_invoff[MAXCH];       // children's inverse relative indices
};
// One front of the fused solve launch (k_solve_tree): everything a workgroup needs in ONE uniform
// load.  The front's factor is stored twice in "solve panel" form, S = [X; -W] with X = inv(L11)
// (unit lower) and W = L21 X, so that each sweep is a single product with no dependency between the
// pivot rows and the update rows (forward: [x^; u] = S f_top + [0; f_below]; backward:
// x = S^T [D^-1 x^; g]):
//   forward copy   thread (row i, column class q):   entries S[i, q + Qf e],        e < Ef
//   backward copy  thread (column k, row class p):   entries S[p + Pb e, k] (rows < w divided by d), e < Eb
// both laid out thread-major (entry e of thread t at e * threads + t): every load instruction of the
// workgroup is one contiguous run, and the entries live in registers across the dependency wait.
struct SolveItem {
  long long spf, spb;        // offsets of the two copies in the solve-panel arenas
  long long uoff, rowoff;    // own update vector; row structure (front rows -> pivot indices)
  int c0, w, r, nchild;
  int Qf, Ef, Pb, Eb;
  long long c_uoff[MAXCH];   // children's update vectors
  int c_invoff[MAXCH];       // children's inverse relative indices (which update row lands on front row i)
  long long Loff;            // the front's panel (source of the solve panels)
  int xbegin, xend;          // children beyond the first MAXCH: entries [xbegin, xend) of the overflow lists
  // A front with more than 1024 rows is several items (slices), each with thread-major copies of ITS rows of S:
  // slice 0 the pivot rows and the update rows [0, a1), slice sl > 0 the update rows [a0, a1).  Forward: every
  // slice forms the front's f_top itself and posts its rows of [x^; u]; backward: the slices sl > 0 post their
  // w partial sums (they only need their ancestors' entries: long before the parent is done), slice 0 polls and
  // adds them in slice order.  One item per front otherwise (a0 = 0, a1 = r - w, nsl = 1).
  int a0, a1, sl, nsl;
  long long poff;            // the front's (nsl - 1) x w partial sums in the partial-sum arena
  int plevel, pad_;          // level of the parent front (-1: a root): a front whose parent lies in the top block of
                             // the solve sends its whole update vector into that block (TopBlockIn::ltop)
};
// Rows of A / columns of K longer than this are not walked by the few lanes of the streaming kernels (16 per row, 8 per
// column: a dense constraint row or a dense Jacobian column would be tens of thousands of dependent iterations of one
// lane group).  They are cut into segments of LONG_SEG entries; a workgroup takes one segment, leaves its partial sum
// in a scratch slot, and the workgroup that arrives last (a counter per row / column) adds the partials IN SEGMENT
// ORDER and finishes the row - deterministic, and a 10^5-entry row is spread over fifty workgroups.
constexpr int LONG_ROW = 1024;
constexpr int LONG_COL = 256;
constexpr int LONG_SEG = 2048;
struct LongSeg {
  int id;         // pivot position of the row / index of the column of K
  int begin, end; // entries [begin, end) of Ar_* / of K
  int idx, nseg;  // this is segment idx of nseg
  int uid;        // counter of the row / column (unique over rows and columns of a plan)
  int poff;       // first partial-sum slot of the row / column
  int pad;
};
// the same for product lists of more than MV_LONG pairs (the diagonal entry of a dense row of A in S = A A^T)
struct LongProd {
  long long e;           // entry of M
  long long begin, end;  // pairs [begin, end)
  int idx, nseg, uid, poff;
};
// ---- the top of the solve tree as ONE dense block (runtime_plan.inc: top block) ----------------------------------
// The last levels of the tree hold a handful of fronts each and cost the solve ~2.7 us per level and direction in
// pure hop latency.  For the fronts T of those levels (closed under "parent", nT columns in all) the forward sweep is
// x^_T = X_T f_T with X_T = inv(L_TT), the backward sweep y_T = X_T^T D_T^-1 x^_T: two dense products, each split
// over many workgroups by rows, two hops instead of two per level.  X_T is formed once per factorisation (lazily:
// the second solve pays for it) from the solve panels of the fronts of T.
struct TopBlockItem {
  long long off;  // this item's thread-major rows in the arena (forward: rows of X_T; backward: rows of X_T^T D^-1)
  int r0, nr;     // rows [r0, r0 + nr) of the block (T-local numbering = pivot order)
  int Q, E;       // column classes, entries per thread (thread (rho, q) owns columns q + Q e, e < E)
};
struct TopFront {
  int item;    // its SolveItem (single-slice fronts only)
  int parent;  // index of the parent's TopFront, -1: root
  int tl0;     // T-local index of its first column
  int pad;
};
struct TopChunk {
  int front;  // TopFront
  int j0;     // first of (at most TOP_CB) columns of that front
};
constexpr int TOP_CB = 4;  // (k_top_inverse: 2 / 4 / 8 / 16 / 32 columns per workgroup measured 156 / 105 / 146 / - / - us, the solve slower with 16 and 32)
// LDS doubles of a k_solve_tree workgroup; a top-block item needs nT + 1024 + (nT + 1 + sources + 1) / 2 of them
constexpr int TOP_LDS = 7 * 1024 + 512;
constexpr int SPMV_NNZ = 2048;  // k_spmv_stream: entries per row block / workgroup
constexpr int SPMV_T = 256;
constexpr int TREE_LDS = 2 * 1024 + 8;  // k_solve_tree without a top block (dynamic LDS, doubles)
struct TopBlockIn {
  int nT, ntf;
  const TopBlockItem* __restrict__ items;  // ntf items
  const double* __restrict__ XTf;          // their thread-major rows of Z = X_T^T D_T^-1 X_T (= inv(S_T)): ONE product y_T = Z f_T
  const int* __restrict__ tpos;       // T-local index -> pivot position
  const int* __restrict__ gptr;       // nT + 1: sources of f_T[i] among the update vectors of the fronts below T ...
  const long long* __restrict__ gsrc; // ... as offsets into uvec (added in this fixed order)
  double* __restrict__ xhatT2;        // 2 x nT exchange slots (by launch parity), sentinel between uses
  double* __restrict__ tT2;           // the same for the rows of the right-hand side t that belong to T ...
  int ntr;                            // ... formed by the FIRST ntr workgroups of the launch (64 rows each; 0: t is in y)
  // how many items of THIS launch have gathered f_T (two counters, by launch parity; item 0 clears
  // the other one).  Every item reads every update vector that reaches T, and a front below T turns around as soon as
  // the rows of y_T it needs are posted - by ONE item, while another may not have gathered yet: the backward items
  // put their update slots back to the sentinel only once all ntf items have read them.
  unsigned int* __restrict__ gathered2;
  int ltop;  // first level of T (only the children of T's fronts feed the block: they are the ones that wait)
};
constexpr int SOLVE_PREFETCH = 32;  // panel entries per thread requested before the dependency wait

constexpr int WIDE_SLICE_ROWS = 256;
constexpr int TOP_REL_CAP = 2048;   // ints of children's relative indices staged in LDS
constexpr int TOP_L21_CAP = 16384;  // doubles of L21 (forward) / inv(L11) (backward) staged in LDS

// info words written by the factorisation kernels
// (INFO_TIMEOUT_WG: 2^30 - the lowest workgroup index of a solve launch whose wait timed out, 0 if none)
enum { INFO_ZERO_PIVOT = 0, INFO_NEG_PIVOT = 1, INFO_TIMEOUT = 2, INFO_TIMEOUT_WG = 3, INFO_WORDS = 4 };
// ... followed by two 64-bit words: pivot minimum (bit-inverted) and maximum (k_pivot_minmax)
constexpr int INFO_BYTES = INFO_WORDS * 4 + 16;

}  // namespace hipfact
