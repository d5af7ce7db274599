// Host-side symbolic analysis for the hipfact KKT backend: structure
// detection, ordering, elimination tree, supernodes, multifrontal maps and the
// level schedule.  Cached per sparsity pattern (the reference's backends redo
// their analysis on every set_matrix: fact_ma57.c:529-625, fact_cholmod.c:133,
// fact_umfpack.c:145-160).
#include "hostmem.h"
#include "plan.h"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <thread>

#include "graph.h"

namespace hipfact {

namespace {

// static-partition parallel loop over [0, n): fn(begin, end, thread_index)
template <class F>
void parallel_chunks(int n, F fn, int grain = 2048) {
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  const int nt = std::max(1, std::min({hw, 32, n / grain + 1}));
  if (nt == 1) {
    fn(0, n, 0);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) {
    const int b = (int)((long long)n * t / nt), e = (int)((long long)n * (t + 1) / nt);
    pool.emplace_back([=, &fn] { fn(b, e, t); });
  }
  for (auto& th : pool) th.join();
}

// the same with the chunks handed out on demand (grain iterations at a time): for loops whose cost per iteration is
// very uneven - a dense constraint row is 10^5 times the work of an ordinary one, and the late rows sit together at
// the end of the pivot order, i.e. in ONE chunk of a static partition
template <class F>
void parallel_dynamic(int n, F fn, int grain = 256) {
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  const int nt = std::max(1, std::min({hw, 32, n / (4 * grain) + 1}));
  if (nt == 1) {
    fn(0, n, 0);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([&, t] {
      for (;;) {
        const int b = next.fetch_add(grain);
        if (b >= n) break;
        fn(b, std::min(n, b + grain), t);
      }
    });
  for (auto& th : pool) th.join();
}

double now_s() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}

// Elimination tree of the matrix whose graph is g, eliminated in order perm
// (Liu's algorithm with path compression).
void etree(const Graph& g, const std::vector<int>& perm, const std::vector<int>& iperm,
           std::vector<int>& parent) {
  const int m = g.n;
  parent.assign(m, -1);
  std::vector<int> anc(m, -1);
  for (int k = 0; k < m; ++k) {
    const int v = perm[k];
    for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
      int r = iperm[g.adj[q]];
      if (r >= k) continue;
      while (anc[r] != -1 && anc[r] != k) {
        const int nx = anc[r];
        anc[r] = k;
        r = nx;
      }
      if (anc[r] == -1) {
        anc[r] = k;
        parent[r] = k;
      }
    }
  }
}

// The same tree for M = A A^T straight from the rows of A (Gilbert, Ng, Peyton: the column elimination tree; the
// form of Davis' cs_etree with ata = 1): rows are visited in elimination order, and a row only has to be linked to the
// PREVIOUS row of each of its columns - nnz(A) find-root walks instead of nnz(A A^T).  Columns flagged in `skip`
// (dense columns, left out of M) do not link anything.
void etree_rows(int m, int nx, const std::vector<int>& ar_ptr, const std::vector<int>& ar_col, const char* skip,
                const std::vector<int>& perm, std::vector<int>& parent) {
  parent.assign(m, -1);
  std::vector<int> anc(m, -1), prev((size_t)nx, -1);
  for (int k = 0; k < m; ++k) {
    const int a = perm[k];
    for (int e = ar_ptr[a]; e < ar_ptr[a + 1]; ++e) {
      const int j = ar_col[e];
      if (skip && skip[j]) continue;
      for (int r = prev[j]; r != -1 && r < k;) {
        const int nxt = anc[r];
        anc[r] = k;
        if (nxt == -1) parent[r] = k;
        r = nxt;
      }
      prev[j] = k;
    }
  }
}

// Postorder of a forest; children visited in ascending key order (so the child
// with the largest key is numbered last, right before its parent).
void postorder(const std::vector<int>& parent, const std::vector<int>* key, std::vector<int>& post) {
  const int m = (int)parent.size();
  std::vector<int> order(m);
  std::iota(order.begin(), order.end(), 0);
  if (key) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (*key)[a] > (*key)[b]; });
  // build child lists by pushing in reverse visiting order
  std::vector<int> head(m, -1), next(m, -1);
  std::vector<int> roots;
  if (!key) {
    for (int k = m - 1; k >= 0; --k) {
      if (parent[k] == -1)
        roots.push_back(k);
      else {
        next[k] = head[parent[k]];
        head[parent[k]] = k;
      }
    }
    std::reverse(roots.begin(), roots.end());
  } else {
    // order is descending by key; pushing front yields ascending lists
    for (int t = 0; t < m; ++t) {
      const int k = order[t];
      if (parent[k] == -1)
        roots.push_back(k);
      else {
        next[k] = head[parent[k]];
        head[parent[k]] = k;
      }
    }
    std::reverse(roots.begin(), roots.end());
  }
  post.clear();
  post.reserve(m);
  std::vector<int> stack;
  for (int r : roots) {
    stack.push_back(r);
    while (!stack.empty()) {
      const int v = stack.back();
      const int c = head[v];
      if (c != -1) {
        head[v] = next[c];
        stack.push_back(c);
      } else {
        stack.pop_back();
        post.push_back(v);
      }
    }
  }
  assert((int)post.size() == m);
}

struct RawSuper {
  int c0 = 0, w = 0;
  std::vector<int> rows;  // sorted, own columns first
  int64_t zeros = 0;
  bool dead = false;
  // columns (relative to c0) at which another subtree joins the chain: the column has further
  // children besides its predecessor.  The structure nests, so the columns form one front, but a
  // front that has to be cut at the width cap is cut THERE: children hang off the part their first
  // update row lies in, and an even cut would push them (and their whole subtree) one level down.
  std::vector<int> joins;
};

// Supernodal symbolic factorisation for a postordered matrix.  Fills the
// supernode partition with row structures and the per-column counts.
//
// One column of the symbolic factorisation (shared by the serial sweep and the per-subtree workers): `sn` is the
// list the column's supernode is appended to / extended in, `mark` a scratch array of m ints private to the caller.
struct SymbolicCtx {
  const Graph& g;
  const std::vector<int>& perm;
  const std::vector<int>& iperm;
  const std::vector<int>& parent;
  const std::vector<int>& head;
  const std::vector<int>& next;
  std::vector<int>& sn_of;      // column -> index of its supernode IN THE LIST IT WAS APPENDED TO
  std::vector<const std::vector<RawSuper>*>& list_of;  // column -> that list
  std::vector<int>& colcount;
};

inline void symbolic_column(const SymbolicCtx& C, int j, std::vector<RawSuper>& sn, std::vector<int>& mark,
                            std::vector<int>& extras) {
  const bool chain = (j > 0 && C.parent[j - 1] == j) && !sn.empty() && sn.back().c0 + sn.back().w == j;
  const int stamp = chain ? sn.back().c0 : j;  // rows of the open supernode are marked with its c0
  extras.clear();
  auto visit = [&](int i) {
    if (i > j && mark[i] != stamp) {
      mark[i] = stamp;
      extras.push_back(i);
    }
  };
  const int v = C.perm[j];
  for (int64_t q = C.g.ptr[v]; q < C.g.ptr[v + 1]; ++q) visit(C.iperm[C.g.adj[q]]);
  for (int c = C.head[j]; c != -1; c = C.next[c]) {
    if (chain && c == j - 1) continue;
    const RawSuper& cs = (*C.list_of[c])[(size_t)C.sn_of[c]];
    // c is the last column of its supernode: struct(c) = below rows
    for (size_t t = cs.w; t < cs.rows.size(); ++t) visit(cs.rows[t]);
  }
  if (chain && extras.empty()) {
    RawSuper& s = sn.back();
    if (C.next[C.head[j]] != -1) s.joins.push_back(j - s.c0);  // more children than the chain predecessor
    s.w += 1;
    C.sn_of[j] = (int)sn.size() - 1;
    C.list_of[j] = &sn;
    C.colcount[j] = (int)s.rows.size() - (j - s.c0);
    return;
  }
  RawSuper ns;
  ns.c0 = j;
  ns.w = 1;
  if (chain) {
    // inherit the tail of the open supernode (it was marked with the old
    // stamp; re-mark with the new one together with the extras)
    const RawSuper& s = sn.back();
    for (size_t t = (size_t)(j - s.c0) + 1; t < s.rows.size(); ++t) extras.push_back(s.rows[t]);
  }
  std::sort(extras.begin(), extras.end());
  ns.rows.reserve(extras.size() + 1);
  ns.rows.push_back(j);
  ns.rows.insert(ns.rows.end(), extras.begin(), extras.end());
  for (int i : ns.rows) mark[i] = j;  // stamp of the new open supernode = its c0
  C.colcount[j] = (int)ns.rows.size();
  C.sn_of[j] = (int)sn.size();
  C.list_of[j] = &sn;
  sn.push_back(std::move(ns));
}

// The columns of a subtree are a contiguous range of a postordered matrix and touch nothing outside it but the row
// indices of their ancestors: disjoint subtrees are processed by threads of their own (private mark arrays, private
// supernode lists), then one serial sweep in column order splices the lists and handles the columns above them
// (the separators at the top of a nested-dissection ordering: few, but with long row lists).
void symbolic(const Graph& g, const std::vector<int>& perm, const std::vector<int>& iperm,
              const std::vector<int>& parent, std::vector<RawSuper>& sn, std::vector<int>& colcount) {
  const int m = g.n;
  sn.clear();
  colcount.assign(m, 0);
  std::vector<int> head(m, -1), next(m, -1);
  for (int k = m - 1; k >= 0; --k)
    if (parent[k] != -1) {
      next[k] = head[parent[k]];
      head[parent[k]] = k;
    }
  std::vector<int> sn_of(m, -1);
  std::vector<const std::vector<RawSuper>*> list_of((size_t)m, nullptr);
  const SymbolicCtx C{g, perm, iperm, parent, head, next, sn_of, list_of, colcount};
  // subtree sizes (children precede parents) and the maximal subtrees of at most `cap` columns
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  const int nt = std::min(hw, 16);
  std::vector<int> first;  // first column of each chosen subtree; root[t] its last
  std::vector<int> root;
  if (nt > 1 && m >= 8192) {
    std::vector<int> size(m, 1);
    for (int k = 0; k < m; ++k)
      if (parent[k] != -1) size[parent[k]] += size[k];
    const int cap = std::max(256, m / (4 * nt));
    for (int k = 0; k < m; ++k) {
      const bool fits = size[k] <= cap;
      const bool parent_fits = parent[k] != -1 && size[parent[k]] <= cap;
      if (fits && !parent_fits && size[k] >= 64) {
        first.push_back(k - size[k] + 1);
        root.push_back(k);
      }
    }
  }
  const int nsub = (int)first.size();
  std::vector<std::vector<RawSuper>> lists((size_t)nsub);
  if (nsub > 0) {
    std::atomic<int> nextsub{0};
    auto worker = [&]() {
      std::vector<int> mark(m, -1), extras;
      for (int t = nextsub++; t < nsub; t = nextsub++)
        for (int j = first[(size_t)t]; j <= root[(size_t)t]; ++j) symbolic_column(C, j, lists[(size_t)t], mark, extras);
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < std::min(nt, nsub); ++t) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
  }
  // serial sweep: splice the lists in column order, process what lies between and above them
  std::vector<int> mark(m, -1), extras;
  int t = 0;
  for (int j = 0; j < m;) {
    if (t < nsub && first[(size_t)t] == j) {
      const int base = (int)sn.size();
      for (RawSuper& s : lists[(size_t)t]) sn.push_back(std::move(s));
      for (int c = j; c <= root[(size_t)t]; ++c) {
        sn_of[c] += base;
        list_of[c] = &sn;
      }
      // the rows of the (possibly still open) last supernode carry its stamp in the worker's mark array, not in this
      // one: re-mark them, a chain column right behind the subtree relies on it
      {
        const RawSuper& s = sn.back();
        for (int i : s.rows) mark[i] = s.c0;
      }
      j = root[(size_t)t] + 1;
      ++t;
      continue;
    }
    symbolic_column(C, j, sn, mark, extras);
    ++j;
  }
  // (sn.reserve() is not used above: list_of holds pointers to the vector OBJECT, which never moves)
}

inline int64_t trapezoid(int64_t w, int64_t r) { return w * r - w * (w - 1) / 2; }

// front that holds column c now (fronts merged into their successor are dead: follow the chain)
inline int col2sn_live(const std::vector<RawSuper>& sn, const std::vector<int>& col2sn, int c) {
  int s = col2sn[c];
  while (s >= 0 && sn[s].dead) ++s;  // a dead front was merged into the next one (adjacent merges only at that point)
  return s;
}

// Returns false when the renumbering of adopted leaves left a row list unsorted (an internal inconsistency that has
// never been observed): the caller repeats the pass without adoption instead of taking the host process down.
bool amalgamate(std::vector<RawSuper>& sn, int m, const PlanParams& prm, std::vector<int>& perm, std::vector<int>& iperm) {
  const int ns = (int)sn.size();
  if (ns == 0) return true;
  std::vector<int> col2sn(m);
  for (int s = 0; s < ns; ++s)
    for (int k = 0; k < sn[s].w; ++k) col2sn[sn[s].c0 + k] = s;
  // children per supernode: a merged front inherits the children of both, and the device keeps
  // fronts with at most max_children children on its fast (pull-mode / single-launch) path
  std::vector<int> nch(ns, 0);
  for (int s = 0; s < ns; ++s)
    if ((int)sn[s].rows.size() > sn[s].w) ++nch[col2sn[sn[s].rows[sn[s].w]]];
  for (int s = 0; s + 1 < ns; ++s) {
    RawSuper& a = sn[s];
    if ((int)a.rows.size() == a.w) continue;  // root
    const int p = col2sn[a.rows[a.w]];
    if (p != s + 1) continue;
    RawSuper& b = sn[p];
    const int wm = a.w + b.w;
    // (a parent that is wider than one front already - a dense leaf clique, cut into a chain by split_wide - takes a
    // small child in as long as the chain gets no longer: the child would be one more LEVEL under it)
    auto parts = [&](int w) { return (w + prm.wmax - 1) / prm.wmax; };
    // ... and a link of a chain (an only child) joins its parent whatever the width: split_wide cuts the merged
    // columns into the fewest fronts that fit, never more than the two had apart
    if (wm > prm.wmax && nch[p] != 1 && (b.w <= prm.wmax || parts(wm) > parts(b.w))) continue;
    if (prm.max_children > 0 && nch[p] - 1 + nch[s] > std::max(prm.max_children, nch[p])) continue;
    const int64_t ua = (int64_t)a.rows.size() - a.w;
    const int64_t rb = (int64_t)b.rows.size();
    const int64_t zeros = a.zeros + b.zeros + (int64_t)a.w * (rb - ua);
    const int64_t tot = trapezoid(wm, a.w + rb);
    const double frac = (double)zeros / (double)tot;
    bool ok;
    if (wm <= 4)
      ok = true;
    else if (wm <= 32)
      ok = frac < prm.relax_small;
    else if (wm <= 64)
      ok = frac < prm.relax_mid;
    else
      ok = frac < prm.relax_big;
    // (a handful of columns never justify a tree level of their own: whatever zeros the parent has collected already)
    ok = ok || (a.w <= 8 && (double)((int64_t)a.w * (rb - ua)) < 0.06 * (double)tot);
    if (!ok) continue;
    std::vector<int> rows;
    rows.reserve(a.w + b.rows.size());
    for (int k = 0; k < a.w; ++k) rows.push_back(a.c0 + k);
    rows.insert(rows.end(), b.rows.begin(), b.rows.end());
    b.rows.swap(rows);
    {
      std::vector<int> joins(a.joins);
      joins.push_back(a.w);
      for (int q : b.joins) joins.push_back(a.w + q);
      b.joins.swap(joins);
    }
    b.c0 = a.c0;
    b.w = wm;
    b.zeros = zeros;
    nch[p] += nch[s] - 1;
    a.dead = true;
    std::vector<int>().swap(a.rows);
  }
  // ---- adoption of childless fronts.  Only the LAST child of a front is adjacent to it in the
  // postorder; the other children stay fronts of their own above, however small - typically the first
  // few vertices a minimum-degree ordering removes from a nearly dense leaf subgraph (one column each,
  // a slightly smaller structure than the clique behind them).  Each of them costs a whole tree level
  // of dependent latency on the device for a handful of explicit zeros.  A childless front may be
  // moved anywhere in front of its parent (any topological order of the elimination tree gives the
  // same fill), so it is renumbered to sit right in front of the parent's columns and merged under
  // the same zero-fraction rule.  Its columns occur in no other front's row list, and all other
  // indices move by a monotone map: the row lists stay sorted.
  std::vector<std::vector<int>> adopted(ns);
  bool any_adopted = false;
  if (prm.adopt_leaves) {
    for (int s = 0; s < ns; ++s) {
      RawSuper& a = sn[s];
      if (a.dead || nch[s] != 0 || (int)a.rows.size() == a.w) continue;
      const int p = col2sn_live(sn, col2sn, a.rows[a.w]);
      if (p == s + 1 || p < 0) continue;
      RawSuper& b = sn[p];
      const int wm = a.w + b.w;
      auto parts = [&](int w) { return (w + prm.wmax - 1) / prm.wmax; };
      if (wm > prm.wmax && (b.w <= prm.wmax || parts(wm) > parts(b.w))) continue;
      const int64_t ua = (int64_t)a.rows.size() - a.w;
      const int64_t rb = (int64_t)b.rows.size();
      const int64_t zeros = a.zeros + b.zeros + (int64_t)a.w * (rb - ua);
      const int64_t tot = trapezoid(wm, a.w + rb);
      const double frac = (double)zeros / (double)tot;
      const bool ok = wm <= 4 || frac < (wm <= 32 ? prm.relax_small : wm <= 64 ? prm.relax_mid : prm.relax_big) ||
                      (a.w <= 8 && (double)((int64_t)a.w * (rb - ua)) < 0.06 * (double)tot);
      if (!ok) continue;
      // b keeps its own numbering for now: rows = [adopted columns | own rows] is formed by the renumbering below
      adopted[p].push_back(s);
      std::vector<int> rows;
      rows.reserve(a.w + b.rows.size());
      rows.insert(rows.end(), a.rows.begin(), a.rows.begin() + a.w);  // (old numbers, own adoptions first; sorted once renumbered)
      rows.insert(rows.end(), b.rows.begin(), b.rows.end());
      b.rows.swap(rows);
      for (int& q : b.joins) q += a.w;
      b.joins.insert(b.joins.begin(), a.w);
      b.w = wm;
      b.zeros = zeros;
      --nch[p];
      a.dead = true;
      std::vector<int>().swap(a.rows);
      any_adopted = true;
    }
  }
  if (any_adopted) {
    // new numbers: fronts in order, the adopted columns of a front right in front of its own
    std::vector<int> newidx(m, -1);
    int next = 0;
    for (int s = 0; s < ns; ++s) {
      if (sn[s].dead) continue;
      RawSuper& b = sn[s];
      const int w = b.w;
      for (int k = 0; k < w; ++k) newidx[b.rows[k]] = next++;  // first w rows = adopted columns, then own columns
    }
    if (next != m) return false;
    for (int s = 0; s < ns; ++s) {
      if (sn[s].dead) continue;
      RawSuper& b = sn[s];
      for (int& r : b.rows) r = newidx[r];
      b.c0 = b.rows[0];
      if (!std::is_sorted(b.rows.begin(), b.rows.end())) return false;  // (never seen; the caller redoes the pass without adoption)
    }
    std::vector<int> perm2(m);
    for (int k = 0; k < m; ++k) perm2[newidx[k]] = perm[k];
    perm.swap(perm2);
    for (int k = 0; k < m; ++k) iperm[perm[k]] = k;
  }
  std::vector<RawSuper> out;
  out.reserve(ns);
  for (auto& s : sn)
    if (!s.dead) out.push_back(std::move(s));
  sn.swap(out);
  return true;
}

void split_wide(std::vector<RawSuper>& sn, int wmax) {
  std::vector<RawSuper> out;
  out.reserve(sn.size());
  for (auto& s : sn) {
    if (s.w <= wmax) {
      out.push_back(std::move(s));
      continue;
    }
    // The fewest parts that fit the cap (every part of the chain is a tree LEVEL on the device), cut evenly; a cut
    // moves to a join of the merged fronts when one lies within an eighth of a part of the even position (in front of
    // a join the columns have the shorter structure they had before the merge) and every part still fits.
    std::vector<int> cuts;  // first columns of the parts
    {
      const int np = (s.w + wmax - 1) / wmax;
      std::vector<int> joins;
      for (int q : s.joins)
        if (q > 0 && q < s.w) joins.push_back(q);
      std::sort(joins.begin(), joins.end());
      cuts.push_back(0);
      for (int p = 1; p < np; ++p) {
        const int ideal = (int)(((long long)p * s.w + np / 2) / np);
        const int lo = std::max(cuts.back() + 1, s.w - (np - p) * wmax), hi = std::min(cuts.back() + wmax, s.w - (np - p));
        int cut = std::min(std::max(ideal, lo), hi);
        const int tol = std::max(1, s.w / (8 * np));
        int best = -1;
        for (int q : joins)
          if (q >= lo && q <= hi && std::abs(q - ideal) <= tol && (best < 0 || std::abs(q - ideal) < std::abs(best - ideal))) best = q;
        if (best >= 0) cut = best;
        cuts.push_back(cut);
      }
    }
    cuts.push_back(s.w);
    for (size_t p = 0; p + 1 < cuts.size(); ++p) {
      RawSuper t;
      t.c0 = s.c0 + cuts[p];
      t.w = cuts[p + 1] - cuts[p];
      t.rows.assign(s.rows.begin() + cuts[p], s.rows.end());
      out.push_back(std::move(t));
    }
  }
  sn.swap(out);
}

}  // namespace

bool build_plan(int N, const int* Kp, const int* Ki, const double* Kx, const PlanParams& prm_in, Plan& P) {
  const double t0 = now_s();
  const bool timing = getenv("HIPFACT_TIMING") != nullptr;
  double tlast = t0;
  auto tick = [&](const char* what) {
    if (timing) {
      const double t = now_s();
      fprintf(stderr, "[hipfact analysis] %-28s %8.2f ms\n", what, (t - tlast) * 1e3);
      tlast = t;
    }
  };
  PlanParams prm = prm_in;
  if (const char* e = getenv("HIPFACT_ORDERING")) prm.ordering = atoi(e);
  if (const char* e = getenv("HIPFACT_ND_LEAF")) prm.nd_leaf = atoi(e);
  if (const char* e = getenv("HIPFACT_WMAX")) prm.wmax = atoi(e);
  if (const char* e = getenv("HIPFACT_MAX_CHILDREN")) prm.max_children = atoi(e);
  if (const char* e = getenv("HIPFACT_ADOPT")) prm.adopt_leaves = atoi(e) != 0;
  if (const char* e = getenv("HIPFACT_DENSE_TAU")) prm.dense_tau = atof(e);
  if (const char* e = getenv("HIPFACT_DENSE_MODE")) prm.dense_mode = atoi(e);
  if (const char* e = getenv("HIPFACT_LATE_TAU")) prm.late_tau = atof(e);
  if (const char* e = getenv("HIPFACT_LATE_MAX")) prm.late_max = atoi(e);
  if (const char* e = getenv("HIPFACT_HUB_TAU")) prm.hub_tau = atof(e);
  if (const char* e = getenv("HIPFACT_PROD_BUDGET")) prm.prod_budget = atof(e);
  if (const char* e = getenv("HIPFACT_ND_SEP_FRAC")) prm.nd_sep_frac = atof(e);
  if (const char* e = getenv("HIPFACT_RELAX")) {
    double a, b, c;
    if (sscanf(e, "%lf,%lf,%lf", &a, &b, &c) == 3) {
      prm.relax_small = a;
      prm.relax_mid = b;
      prm.relax_big = c;
    }
  }
  if (prm.wmax < 1) prm.wmax = 1;
  if (prm.wmax > 128) prm.wmax = 128;

  P = Plan();
  P.N = N;
  if (N < 0 || Kp == nullptr) {
    P.error = "invalid matrix";
    return false;
  }
  const int64_t nnz = N > 0 ? Kp[N] : 0;
  P.nnzK = nnz;
  // ---- validate: lower triangular, strictly ascending rows
  for (int j = 0; j < N; ++j) {
    if (Kp[j + 1] < Kp[j]) {
      P.error = "column pointers not monotone";
      return false;
    }
    for (int e = Kp[j]; e < Kp[j + 1]; ++e) {
      if (Ki[e] < j || Ki[e] >= N) {
        P.error = "matrix is not lower triangular (SLEQP_FACT_FLAGS_LOWER expected)";
        return false;
      }
      if (e > Kp[j] && Ki[e] <= Ki[e - 1]) {
        P.error = "row indices not strictly ascending";
        return false;
      }
    }
  }
  P.Kp.assign(Kp, Kp + N + 1);
  P.Ki.assign(Ki, Ki + nnz);

  // ---- structure detection: K = [I A^T; A 0] with empty trailing columns
  int n = N;
  while (n > 0 && Kp[n] == Kp[n - 1]) --n;
  bool shape = !prm.force_generic;
  if (n == 0 && N > 0) shape = false;  // all-zero matrix
  for (int j = 0; j < n && shape; ++j) {
    const int e0 = Kp[j];
    if (Kp[j + 1] == e0 || Ki[e0] != j) shape = false;
    else if (Kp[j + 1] > e0 + 1 && Ki[e0 + 1] < n) shape = false;
  }
  bool saddle = shape;
  for (int j = 0; j < n && saddle; ++j)
    if (Kx && Kx[Kp[j]] != 1.0) saddle = false;
  P.saddle_shape = shape;
  P.n_shape = shape ? n : 0;
  P.saddle = saddle;
  P.n = saddle ? n : 0;
  P.m = saddle ? N - n : N;
  const int my = P.m;  // constraint rows (saddle) / order of the matrix (generic)
  const int nx = P.n;
  P.my = saddle ? my : 0;

  // ---- dense columns of A
  // mode 2 (low-rank correction, dense_cols.inc): left out of S (their rows would form cliques); handled by the solves
  std::vector<char> is_dense;
  if (saddle && prm.dense_mode == 2 && prm.dense_tau > 0.0 && prm.dense_max > 0 && my > 0) {
    const double thr = std::max((double)prm.dense_min, prm.dense_tau * std::sqrt((double)my));
    std::vector<std::pair<int, int>> cand;  // (-count, column)
    for (int j = 0; j < nx; ++j) {
      const int c = Kp[j + 1] - Kp[j] - 1;
      if ((double)c > thr) cand.push_back({-c, j});
    }
    if (!cand.empty()) {
      std::sort(cand.begin(), cand.end());
      if ((int)cand.size() > prm.dense_max) cand.resize((size_t)prm.dense_max);
      is_dense.assign((size_t)nx, 0);
      for (auto& c : cand) {
        is_dense[(size_t)c.second] = 1;
        P.dense_cols.push_back(c.second);
      }
      std::sort(P.dense_cols.begin(), P.dense_cols.end());
    }
  }
  // mode 1 (late elimination): the variable stays a vertex of the graph, ordered behind the constraint rows.  Chosen
  // by count (the clique c^2 / 2 against the at most m entries of a late row of L) and by the budget of the product
  // lists (sum over the ordinary columns of c (c + 1) / 2 pairs).
  std::vector<int> late_of;  // per x column: index of its late vertex, -1 = ordinary
  int k_late = 0;
  if (saddle && prm.dense_mode == 1 && my > 0) {
    const double thr = std::max((double)prm.late_min, prm.late_tau * std::sqrt((double)my));
    const int cap = prm.late_max > 0 ? prm.late_max : std::max(64, my / 4);
    double pairs = 0.0;
    std::vector<std::pair<int, int>> cand;  // (-count, column): everything that could matter for the budget
    for (int j = 0; j < nx; ++j) {
      const int c = Kp[j + 1] - Kp[j] - 1;
      pairs += 0.5 * (double)c * (c + 1);
      if (c >= 16) cand.push_back({-c, j});
    }
    std::sort(cand.begin(), cand.end());
    for (const auto& cj : cand) {
      const int c = -cj.first;
      if (k_late >= cap || !((double)c > thr || pairs > prm.prod_budget)) break;
      if (late_of.empty()) late_of.assign((size_t)nx, -1);
      late_of[(size_t)cj.second] = 0;  // (numbered below, in column order)
      pairs -= 0.5 * (double)c * (c + 1);
      ++k_late;
    }
    if (k_late > 0) {
      int t = 0;
      for (int j = 0; j < nx; ++j)
        if (late_of[(size_t)j] >= 0) {
          late_of[(size_t)j] = t++;
          P.late_cols.push_back(j);
        }
    }
    if (pairs > 2.0e9) {
      P.error = "product list too large (the constraint Jacobian has too many dense columns for S = A A^T)";
      return false;
    }
  }
  P.n_late = k_late;
  const int m = my + k_late;  // order of M
  P.m = m;
  // columns that do not enter the products of S = A_s A_s^T (either mode)
  std::vector<char> skipcol;
  if (!is_dense.empty())
    skipcol = is_dense;
  else if (k_late > 0) {
    skipcol.assign((size_t)nx, 0);
    for (int j : P.late_cols) skipcol[(size_t)j] = 1;
  }
  const char* dn = skipcol.empty() ? nullptr : skipcol.data();
  const int* lateof = late_of.empty() ? nullptr : late_of.data();
  // ---- graph of M (saddle: S = A_s A_s^T plus the late variables' rows, generic: K + K^T), no diagonal
  Graph g;
  g.n = m;
  g.ptr.assign(m + 1, 0);
  std::vector<int> ar_ptr, ar_col, ar_src;  // CSR of A in original row order
  std::vector<char> late_row;               // constraint rows ordered behind the late variables
  int n_late_rows = 0;
  if (saddle) {
    ar_ptr.assign(my + 1, 0);
    for (int j = 0; j < nx; ++j)
      for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e) ++ar_ptr[Ki[e] - nx + 1];
    for (int a = 0; a < my; ++a) ar_ptr[a + 1] += ar_ptr[a];
    ar_col.resize(ar_ptr[my]);
    ar_src.resize(ar_ptr[my]);
    std::vector<int> fill(ar_ptr.begin(), ar_ptr.end() - 1);
    for (int j = 0; j < nx; ++j)
      for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e) {
        const int a = Ki[e] - nx;
        ar_col[fill[a]] = j;
        ar_src[fill[a]] = e;
        ++fill[a];
      }
    // A row whose ONLY entry lies in a dense column (the unit row of an active bound on that variable, when K's own
    // pattern is analysed) would vanish from A_s: that column stays an ordinary one (x_j is pinned by the row)
    if (!is_dense.empty()) {
      // (mode 2; ADVICE round 3: not only singleton rows - every row has to keep an entry outside the dense columns,
      // or S_s gets a zero row although K is regular)
      bool changed = false;
      for (int a = 0; a < my; ++a) {
        bool has_ordinary = ar_ptr[a + 1] == ar_ptr[a];
        for (int q = ar_ptr[a]; q < ar_ptr[a + 1] && !has_ordinary; ++q) has_ordinary = !is_dense[(size_t)ar_col[q]];
        if (!has_ordinary) {
          is_dense[(size_t)ar_col[ar_ptr[a]]] = 0;
          changed = true;
        }
      }
      if (changed) {
        P.dense_cols.clear();
        for (int j = 0; j < nx; ++j)
          if (is_dense[(size_t)j]) P.dense_cols.push_back(j);
        skipcol = is_dense;
        dn = skipcol.data();
      }
    }
    // mode 1: such a row is ordered BEHIND the late variables instead (its pivot then is the positive
    // sum of a_id^2 / |d_d| and more; in front of them it would be an exact zero)
    if (k_late > 0) {
      late_row.assign((size_t)my, 0);
      for (int a = 0; a < my; ++a) {
        bool has_ordinary = ar_ptr[a + 1] == ar_ptr[a];
        for (int q = ar_ptr[a]; q < ar_ptr[a + 1] && !has_ordinary; ++q) has_ordinary = !skipcol[(size_t)ar_col[q]];
        if (!has_ordinary) {
          late_row[(size_t)a] = 1;
          ++n_late_rows;
        }
      }
    }
    // rows are independent (count, prefix, fill; one marker array per thread).  Row a < my: the rows that share an
    // ordinary column with it and the late variables among its columns; vertex my + t: the rows of late column t.
    std::vector<int64_t> deg(m + 1, 0);
    std::vector<std::vector<int>> mark_of((size_t)32);  // (one marker array per thread, kept across its chunks)
    parallel_dynamic(my, [&](int lo, int hi, int tid) {
      std::vector<int>& mark = mark_of[(size_t)tid];
      if (mark.empty()) mark.assign((size_t)my, -1);
      for (int a = lo; a < hi; ++a) {
        mark[a] = a;
        int64_t c = 0;
        for (int q = ar_ptr[a]; q < ar_ptr[a + 1]; ++q) {
          const int j = ar_col[q];
          if (dn && dn[j]) {
            if (lateof && lateof[j] >= 0) ++c;
            continue;
          }
          for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e) {
            const int b = Ki[e] - nx;
            if (mark[b] != a) {
              mark[b] = a;
              ++c;
            }
          }
        }
        deg[a + 1] = c;
      }
    });
    for (int t = 0; t < k_late; ++t) deg[my + t + 1] = Kp[P.late_cols[(size_t)t] + 1] - Kp[P.late_cols[(size_t)t]] - 1;
    for (int a = 0; a < m; ++a) deg[a + 1] += deg[a];
    g.ptr.assign(deg.begin(), deg.end());
    huge_resize(g.adj, (size_t)g.ptr[m]);
    for (auto& mk : mark_of) std::fill(mk.begin(), mk.end(), -1);
    parallel_dynamic(my, [&](int lo, int hi, int tid) {
      std::vector<int>& mark = mark_of[(size_t)tid];
      if (mark.empty()) mark.assign((size_t)my, -1);
      for (int a = lo; a < hi; ++a) {
        mark[a] = a;
        int64_t o = g.ptr[a];
        for (int q = ar_ptr[a]; q < ar_ptr[a + 1]; ++q) {
          const int j = ar_col[q];
          if (dn && dn[j]) {
            if (lateof && lateof[j] >= 0) g.adj[o++] = my + lateof[j];
            continue;
          }
          for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e) {
            const int b = Ki[e] - nx;
            if (mark[b] != a) {
              mark[b] = a;
              g.adj[o++] = b;
            }
          }
        }
      }
    });
    for (int t = 0; t < k_late; ++t) {
      const int d = P.late_cols[(size_t)t];
      int64_t o = g.ptr[my + t];
      for (int e = Kp[d] + 1; e < Kp[d + 1]; ++e) g.adj[o++] = Ki[e] - nx;
    }
    // hub rows: adjacent to a large share of all rows - out of the dissection, ordered last
    if (prm.hub_tau > 0.0 && prm.ordering != 2) {
      const double thr = std::max((double)prm.hub_min, prm.hub_tau * std::sqrt((double)my));
      std::vector<std::pair<int64_t, int>> cand;  // (-degree, row)
      for (int a = 0; a < my; ++a)
        if ((double)(g.ptr[a + 1] - g.ptr[a]) > thr) cand.push_back({-(g.ptr[a + 1] - g.ptr[a]), a});
      // at most max(64, m / 8) of them, the best connected first (when a large share of all rows qualifies the graph
      // is simply dense: taking the worst of them out still shortens the separators of the rest)
      const size_t cap = (size_t)std::max(64, my / 8);
      if (cand.size() > cap) {
        std::partial_sort(cand.begin(), cand.begin() + (long)cap, cand.end());
        cand.resize(cap);
      }
      if (!cand.empty()) {
        if (late_row.empty()) late_row.assign((size_t)my, 0);
        for (const auto& c : cand)
          if (!late_row[(size_t)c.second]) {
            late_row[(size_t)c.second] = 1;
            ++n_late_rows;
          }
      }
    }
    if (n_late_rows == 0) late_row.clear();
  } else {
    std::vector<int64_t> cnt(m + 1, 0);
    for (int j = 0; j < N; ++j)
      for (int e = Kp[j]; e < Kp[j + 1]; ++e)
        if (Ki[e] != j) {
          ++cnt[j + 1];
          ++cnt[Ki[e] + 1];
        }
    for (int a = 0; a < m; ++a) cnt[a + 1] += cnt[a];
    g.ptr.assign(cnt.begin(), cnt.end());
    huge_resize(g.adj, (size_t)cnt[m]);
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    for (int j = 0; j < N; ++j)
      for (int e = Kp[j]; e < Kp[j + 1]; ++e)
        if (Ki[e] != j) {
          g.adj[fill[j]++] = Ki[e];
          g.adj[fill[Ki[e]]++] = j;
        }
  }
  P.n_late_rows = n_late_rows;

  tick("validate + graph of M");
  // ---- ordering
  const double t1 = now_s();
  std::vector<int> perm;
  auto order_graph = [&](const Graph& gg, std::vector<int>& pp) {
    if (prm.ordering == 2) {
      pp.resize((size_t)gg.n);
      std::iota(pp.begin(), pp.end(), 0);
    } else if (prm.ordering == 1) {
      amd_order(gg, pp);
    } else {
      NDParams nd;
      nd.max_sep_frac = prm.nd_sep_frac;
      if (prm.nd_leaf > 0)
        nd.leaf_size = prm.nd_leaf;
      else
        nd.leaf_size = std::max(prm.wmax, 32);  // a leaf subgraph that fits one front is not dissected further
      if (const char* e = getenv("HIPFACT_ND_SMALL_SEP_FRAC")) nd.small_sep_frac = atof(e);
      if (const char* e = getenv("HIPFACT_ND_REFINE")) nd.refine = atoi(e) != 0;
      if (const char* e = getenv("HIPFACT_ND_BALANCE")) nd.balance = atof(e);
      if (const char* e = getenv("HIPFACT_ND_BALANCE_W")) nd.balance_weight = atof(e);
      if (const char* e = getenv("HIPFACT_ND_DEPTH_TOL")) nd.depth_tol = atof(e);
      nd_order(gg, nd, pp);
    }
  };
  if (k_late == 0 && n_late_rows == 0) {
    order_graph(g, perm);
  } else {
    // the ordinary rows are ordered on their own graph; behind them the late variables, then the late rows
    std::vector<int> newid((size_t)m, -1), oldid;
    oldid.reserve((size_t)my);
    for (int a = 0; a < my; ++a)
      if (late_row.empty() || !late_row[(size_t)a]) {
        newid[(size_t)a] = (int)oldid.size();
        oldid.push_back(a);
      }
    const int mr = (int)oldid.size();
    Graph gr;
    gr.n = mr;
    gr.ptr.assign((size_t)mr + 1, 0);
    parallel_chunks(mr, [&](int lo, int hi, int) {
      for (int r = lo; r < hi; ++r) {
        const int a = oldid[(size_t)r];
        int64_t c = 0;
        for (int64_t q = g.ptr[a]; q < g.ptr[a + 1]; ++q) c += newid[(size_t)g.adj[q]] >= 0;
        gr.ptr[(size_t)r + 1] = c;
      }
    });
    for (int r = 0; r < mr; ++r) gr.ptr[(size_t)r + 1] += gr.ptr[(size_t)r];
    huge_resize(gr.adj, (size_t)gr.ptr[(size_t)mr]);
    parallel_chunks(mr, [&](int lo, int hi, int) {
      for (int r = lo; r < hi; ++r) {
        const int a = oldid[(size_t)r];
        int64_t o = gr.ptr[(size_t)r];
        for (int64_t q = g.ptr[a]; q < g.ptr[a + 1]; ++q) {
          const int b = newid[(size_t)g.adj[q]];
          if (b >= 0) gr.adj[o++] = b;
        }
      }
    });
    std::vector<int> pr;
    order_graph(gr, pr);
    perm.resize((size_t)m);
    int k = 0;
    for (int r = 0; r < mr; ++r) perm[(size_t)k++] = oldid[(size_t)pr[(size_t)r]];
    for (int t = 0; t < k_late; ++t) perm[(size_t)k++] = my + t;
    for (int a = 0; a < my; ++a)
      if (!late_row.empty() && late_row[(size_t)a]) perm[(size_t)k++] = a;
  }
  std::vector<int> iperm(m);
  for (int k = 0; k < m; ++k) iperm[perm[k]] = k;
  P.t_order = now_s() - t1;
  tick("ordering");

  // ---- elimination tree, postorder (twice: second time with children sorted
  // by column count so that the heaviest child is adjacent to its parent)
  const double t2 = now_s();
  std::vector<int> parent, post, colcount;
  std::vector<RawSuper> sn;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) {
      if (saddle && k_late == 0 && !getenv("HIPFACT_ETREE_GRAPH"))
        etree_rows(m, nx, ar_ptr, ar_col, dn, perm, parent);
      else
        etree(g, perm, iperm, parent);
      tick("  etree");
    }
    postorder(parent, pass == 0 ? nullptr : &colcount, post);
    std::vector<int> perm2(m), inv(m), parent2(m);
    for (int k = 0; k < m; ++k) {
      perm2[k] = perm[post[k]];
      inv[post[k]] = k;
    }
    for (int k = 0; k < m; ++k) parent2[inv[k]] = parent[k] < 0 ? -1 : inv[parent[k]];
    perm.swap(perm2);
    parent.swap(parent2);
    for (int k = 0; k < m; ++k) iperm[perm[k]] = k;
    tick("  postorder + relabel");
    if (pass == 0 && !getenv("HIPFACT_SYMBOLIC_KEY")) {
      // The key of the second postorder (heaviest child last, i.e. next to its parent): the size of the child's
      // subtree instead of its column count - no symbolic pass for the key alone (config 4: analysis 81 -> 75 ms on
      // the bench host, 595 fronts instead of 600 on the same 11 levels, the solve 2.5 us faster;
      // HIPFACT_SYMBOLIC_KEY=1: the column counts of a first symbolic pass, as in rounds 1-3)
      colcount.assign((size_t)m, 1);
      for (int k = 0; k < m; ++k)
        if (parent[k] >= 0) colcount[(size_t)parent[k]] += colcount[(size_t)k];
      continue;
    }
    symbolic(g, perm, iperm, parent, sn, colcount);
    tick("  symbolic");
  }
  P.nnzL_true = 0;
  P.flops = 0;
  for (int k = 0; k < m; ++k) {
    P.nnzL_true += colcount[k];
    P.flops += (double)colcount[k] * colcount[k];
  }
  tick("etree + symbolic (2 passes)");
  if (!amalgamate(sn, m, prm, perm, iperm)) {
    // (a failed pass has not touched perm / iperm yet; the fronts are rebuilt and merged without adoption)
    PlanParams plain = prm;
    plain.adopt_leaves = false;
    symbolic(g, perm, iperm, parent, sn, colcount);
    if (!prm.adopt_leaves || !amalgamate(sn, m, plain, perm, iperm)) {
      P.error = "internal: amalgamation left an unsorted row structure";
      return false;
    }
  }
  split_wide(sn, prm.wmax);

  // ---- final supernode arrays
  const int ns = (int)sn.size();
  P.nsuper = ns;
  P.perm = perm;
  P.iperm = iperm;
  P.sn_c0.resize(ns + 1);
  P.sn_r.resize(ns);
  P.sn_rowptr.assign(ns + 1, 0);
  P.sn_parent.assign(ns, -1);
  P.sn_level.assign(ns, 0);
  P.sn_Loff.resize(ns);
  P.sn_Uoff.resize(ns);
  P.sn_uoff.resize(ns);
  P.rel_ptr.assign(ns + 1, 0);
  std::vector<int> col2sn(m);
  for (int s = 0; s < ns; ++s) {
    P.sn_c0[s] = sn[s].c0;
    P.sn_r[s] = (int)sn[s].rows.size();
    P.sn_rowptr[s + 1] = P.sn_rowptr[s] + P.sn_r[s];
    for (int k = 0; k < sn[s].w; ++k) col2sn[sn[s].c0 + k] = s;
  }
  P.sn_c0[ns] = m;
  P.sn_rows.resize(P.sn_rowptr[ns]);
  int64_t Loff = 0, Uoff = 0, uoff = 0;
  for (int s = 0; s < ns; ++s) {
    std::copy(sn[s].rows.begin(), sn[s].rows.end(), P.sn_rows.begin() + P.sn_rowptr[s]);
    const int w = sn[s].w, r = P.sn_r[s], u = r - w;
    P.sn_Uoff[s] = Uoff;
    P.sn_uoff[s] = uoff;
    // keep every panel / update matrix 16-byte aligned (even number of doubles)
    Uoff += ((int64_t)u * u + 1) & ~1LL;
    uoff += (u + 1) & ~1;
    P.rel_ptr[s + 1] = P.rel_ptr[s] + u;
    if (u > 0) P.sn_parent[s] = col2sn[sn[s].rows[w]];
    P.nnzL += trapezoid(w, r);
    for (int k = 0; k < w; ++k) P.flops_dense += (double)(r - k) * (r - k);
    P.max_r = std::max(P.max_r, r);
    P.max_w = std::max(P.max_w, w);
    P.max_u = std::max(P.max_u, u);
  }
  P.U_size = Uoff;
  P.u_size = uoff;
  // children, levels
  P.child_ptr.assign(ns + 1, 0);
  for (int s = 0; s < ns; ++s)
    if (P.sn_parent[s] >= 0) ++P.child_ptr[P.sn_parent[s] + 1];
  for (int s = 0; s < ns; ++s) P.child_ptr[s + 1] += P.child_ptr[s];
  P.child_idx.resize(P.child_ptr[ns]);
  {
    std::vector<int> fill(P.child_ptr.begin(), P.child_ptr.end() - 1);
    for (int s = 0; s < ns; ++s)
      if (P.sn_parent[s] >= 0) P.child_idx[fill[P.sn_parent[s]]++] = s;
  }
  int nlev = 0;
  for (int s = 0; s < ns; ++s) {  // children precede parents
    const int p = P.sn_parent[s];
    if (p >= 0) P.sn_level[p] = std::max(P.sn_level[p], P.sn_level[s] + 1);
    nlev = std::max(nlev, P.sn_level[s] + 1);
  }
  P.nlevels = nlev;
  P.level_ptr.assign(nlev + 1, 0);
  for (int s = 0; s < ns; ++s) ++P.level_ptr[P.sn_level[s] + 1];
  for (int l = 0; l < nlev; ++l) P.level_ptr[l + 1] += P.level_ptr[l];
  P.level_sn.resize(ns);
  {
    std::vector<int> fill(P.level_ptr.begin(), P.level_ptr.end() - 1);
    for (int s = 0; s < ns; ++s) P.level_sn[fill[P.sn_level[s]]++] = s;
    // heaviest fronts first inside a level (better tail behaviour)
    for (int l = 0; l < nlev; ++l)
      std::stable_sort(P.level_sn.begin() + P.level_ptr[l], P.level_sn.begin() + P.level_ptr[l + 1],
                       [&](int a, int b) {
                         const double wa = (double)P.sn_r[a] * P.sn_r[a] * (P.sn_c0[a + 1] - P.sn_c0[a]);
                         const double wb = (double)P.sn_r[b] * P.sn_r[b] * (P.sn_c0[b + 1] - P.sn_c0[b]);
                         return wa > wb;
                       });
  }
  // Panels in LEVEL order: the fronts below any level form a prefix of the factor arena.  (The solve-panel items of
  // the device runtime put the panels of the finished bottom levels back to zero behind them, so that the zero fill
  // in front of the next factorisation covers the top levels only.)
  const bool loff_post = getenv("HIPFACT_LOFF_POSTORDER") != nullptr;
  for (int q = 0; q < ns; ++q) {
    const int s = loff_post ? q : P.level_sn[q];
    const int w = sn[s].w, r = P.sn_r[s];
    P.sn_Loff[s] = Loff;
    Loff += ((int64_t)r * w + 1) & ~1LL;
  }
  P.L_size = Loff;
  // Update-matrix arena with reuse.  The update matrix of a front is read by the workgroups of its PARENT only
  // (pull extend-add), and the parent's own update matrix is written after that - so a front may take over the slot
  // of any proper descendant two or more generations below it: everything that reads such a slot is itself a
  // descendant of one of the front's children, which the front awaits before its first workgroup touches memory
  // (per-level launches: kernel boundaries; dataflow launch: the children's Schur counters).  Every front hands its
  // parent the slots of its own children plus whatever it was handed and did not use; a front takes the smallest
  // slot that fits, or fresh memory.  A chain of fronts (dense Schur complement) ping-pongs between two slots:
  // the arena shrinks from sum(u^2) - (m / 128) m^2 / 3 doubles for a dense S of order m - to the two largest.
  if (prm.reuse_update_arena && !getenv("HIPFACT_U_NOREUSE")) {
    struct Slot {
      int64_t off, size;
    };
    std::vector<std::vector<Slot>> handed((size_t)ns);
    int64_t top = 0;
    for (int s = 0; s < ns; ++s) {  // children precede parents
      const int64_t u = P.sn_r[s] - sn[s].w, need = (u * u + 1) & ~1LL;
      std::vector<Slot> cand;
      for (int ci = P.child_ptr[s]; ci < P.child_ptr[s + 1]; ++ci) {
        std::vector<Slot>& hc = handed[(size_t)P.child_idx[ci]];
        cand.insert(cand.end(), hc.begin(), hc.end());
        std::vector<Slot>().swap(hc);
      }
      // neighbours in memory become one slot (siblings were carved out of fresh memory side by side: a parent whose
      // update matrix is bigger than any of its grandchildren's still fits into several of them together)
      std::sort(cand.begin(), cand.end(), [](const Slot& a, const Slot& b) { return a.off < b.off; });
      {
        size_t o = 0;
        for (size_t k = 0; k < cand.size(); ++k) {
          if (o > 0 && cand[o - 1].off + cand[o - 1].size == cand[k].off)
            cand[o - 1].size += cand[k].size;
          else
            cand[o++] = cand[k];
        }
        cand.resize(o);
      }
      int best = -1;
      if (need > 0)
        for (int k = 0; k < (int)cand.size(); ++k)
          if (cand[k].size >= need && (best < 0 || cand[k].size < cand[best].size)) best = k;
      if (need == 0) {
        P.sn_Uoff[s] = 0;
      } else if (best >= 0) {
        P.sn_Uoff[s] = cand[best].off;
        cand[best].off += need;  // the rest of the slot stays available
        cand[best].size -= need;
        if (cand[best].size == 0) cand.erase(cand.begin() + best);
      } else {
        P.sn_Uoff[s] = top;
        top += need;
      }
      for (int ci = P.child_ptr[s]; ci < P.child_ptr[s + 1]; ++ci) {
        const int c = P.child_idx[ci];
        const int64_t uc = P.sn_r[c] - sn[c].w, sz = (uc * uc + 1) & ~1LL;
        if (sz > 0) cand.push_back({P.sn_Uoff[c], sz});
      }
      // (a handful of the largest is all an ancestor can use)
      std::sort(cand.begin(), cand.end(), [](const Slot& a, const Slot& b) { return a.size > b.size; });
      if (cand.size() > 8) cand.resize(8);
      handed[(size_t)s] = std::move(cand);
    }
    P.U_size = top;
  }
  // relative indices: position of each below-row in the parent's front
  P.rel.resize(P.rel_ptr[ns]);
  for (int s = 0; s < ns; ++s) {
    const int p = P.sn_parent[s];
    if (p < 0) continue;
    const int w = sn[s].w;
    const std::vector<int>& rs = sn[s].rows;
    const std::vector<int>& rp = sn[p].rows;
    size_t t = 0;
    for (size_t q = w; q < rs.size(); ++q) {
      while (t < rp.size() && rp[t] < rs[q]) ++t;
      if (t == rp.size() || rp[t] != rs[q]) {
        P.error = "internal: child structure not contained in parent";
        return false;
      }
      P.rel[P.rel_ptr[s] + (q - w)] = (int)t;
    }
  }

  tick("supernode arrays + rel");
  // ---- M pattern in pivot order with target offsets into the L arena
  P.Mp.assign(m + 1, 0);
  parallel_chunks(m, [&](int lo, int hi, int) {
    for (int k = lo; k < hi; ++k) {
      const int v = perm[k];
      int64_t c = 1;
      for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q)
        if (iperm[g.adj[q]] > k) ++c;
      P.Mp[k + 1] = c;
    }
  });
  for (int k = 0; k < m; ++k) P.Mp[k + 1] += P.Mp[k];
  huge_resize(P.Mi, (size_t)P.Mp[m]);
  huge_resize(P.Mtarget, (size_t)P.Mp[m]);
  {
    std::atomic<bool> bad{false};
    std::vector<std::vector<int>> pos_sn((size_t)32);
    parallel_dynamic(ns, [&](int lo, int hi, int tid) {
      std::vector<int>& pos = pos_sn[(size_t)tid];
      if (pos.empty()) pos.assign((size_t)m, -1);
      for (int s = lo; s < hi; ++s) {
        const std::vector<int>& rs = sn[s].rows;
        const int r = (int)rs.size();
        for (int t = 0; t < r; ++t) pos[rs[t]] = t;
        for (int k = sn[s].c0; k < sn[s].c0 + sn[s].w; ++k) {
          const int v = perm[k];
          int64_t o = P.Mp[k];
          P.Mi[o++] = k;
          for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
            const int i = iperm[g.adj[q]];
            if (i > k) P.Mi[o++] = i;
          }
          std::sort(P.Mi.begin() + P.Mp[k] + 1, P.Mi.begin() + P.Mp[k + 1]);
          for (int64_t e = P.Mp[k]; e < P.Mp[k + 1]; ++e) {
            const int t = pos[P.Mi[e]];
            if (t < 0) {
              bad = true;
              continue;
            }
            P.Mtarget[e] = P.sn_Loff[s] + t + (int64_t)(k - sn[s].c0) * r;
          }
        }
        for (int t = 0; t < r; ++t) pos[rs[t]] = -1;
      }
    }, 8);
    if (bad) {
      P.error = "internal: matrix entry outside front";
      return false;
    }
  }

  tick("M pattern + targets");
  // ---- value sources
  if (saddle) {
    // product lists: M(i,k) = sum_j A(perm[i], j) A(perm[k], j)
    huge_resize(P.prod_ptr, (size_t)P.Mp[m] + 1);
    if (P.Mp[m] + 1 < (int64_t)INT32_MAX)
      parallel_chunks((int)(P.Mp[m] + 1), [&](int lo, int hi, int) {
        std::fill(P.prod_ptr.begin() + lo, P.prod_ptr.begin() + hi, (int64_t)0);
      });
    else
      std::fill(P.prod_ptr.begin(), P.prod_ptr.end(), (int64_t)0);
    // pass 0 counts, pass 1 fills; columns are independent (one pos array per thread)
    BigVec<int64_t> fill;
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) {
        tick("  products counted");
        for (int64_t e = 0; e < P.Mp[m]; ++e) P.prod_ptr[e + 1] += P.prod_ptr[e];
        P.nprod = P.prod_ptr[P.Mp[m]];
        if (P.nprod > (int64_t)1 << 31) {
          P.error = "product list too large (dense column in the constraint Jacobian?)";
          return false;
        }
        huge_resize(P.prod_a, (size_t)P.nprod);
        huge_resize(P.prod_b, (size_t)P.nprod);
        huge_resize(fill, P.prod_ptr.size() - 1);
        if (fill.size() < (size_t)INT32_MAX)
          parallel_chunks((int)fill.size(), [&](int lo, int hi, int) {
            std::copy(P.prod_ptr.begin() + lo, P.prod_ptr.begin() + hi, fill.begin() + lo);
          });
        else
          std::copy(P.prod_ptr.begin(), P.prod_ptr.end() - 1, fill.begin());
      }
      std::vector<std::vector<int>> pos_of((size_t)32);
      parallel_dynamic(m, [&](int lo, int hi, int tid) {
        std::vector<int>& pos = pos_of[(size_t)tid];
        if (pos.empty()) pos.assign((size_t)m, -1);
        for (int k = lo; k < hi; ++k) {
          for (int64_t e = P.Mp[k]; e < P.Mp[k + 1]; ++e) pos[P.Mi[e]] = (int)(e - P.Mp[k]);
          const int b = perm[k];
          auto one_pair = [&](int i, int ea, int eb) {  // a single product: an entry between a row and a late variable
            const int64_t me = P.Mp[k] + pos[i];
            if (pass == 0)
              ++P.prod_ptr[me + 1];
            else {
              P.prod_a[fill[me]] = ea;
              P.prod_b[fill[me]] = eb;
              ++fill[me];
            }
          };
          if (b >= my) {
            // late variable x_d (its diagonal has no product: the numeric phase sets it to -1): the entries
            // a^_id = K(e) * 1 (the unit diagonal of column d) towards the rows of its column ordered behind it
            const int d = P.late_cols[(size_t)(b - my)];
            for (int e = Kp[d] + 1; e < Kp[d + 1]; ++e) {
              const int i = iperm[Ki[e] - nx];
              if (i > k) one_pair(i, e, Kp[d]);
            }
            continue;
          }
          for (int q = ar_ptr[b]; q < ar_ptr[b + 1]; ++q) {
            const int j = ar_col[q];
            if (dn && dn[j]) {
              if (lateof && lateof[j] >= 0) {  // ... and towards the late variables ordered behind this row
                const int i = iperm[my + lateof[j]];
                if (i > k) one_pair(i, ar_src[q], Kp[j]);
              }
              continue;
            }
            const int eb = ar_src[q];
            for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e) {
              const int i = iperm[Ki[e] - nx];
              if (i < k) continue;
              const int64_t me = P.Mp[k] + pos[i];
              if (pass == 0)
                ++P.prod_ptr[me + 1];
              else {
                P.prod_a[fill[me]] = e;
                P.prod_b[fill[me]] = eb;
                ++fill[me];
              }
            }
          }
        }
      });
    }
    tick("  products filled");
    // SpMV structures
    // (the row of a late variable x_d in this CSR: its own unit entry, so that the right-hand side of its
    // equation -x_d' + A_d^T y = b_d is b_d)
    P.Ar_ptr.assign(m + 1, 0);
    for (int k = 0; k < m; ++k)
      P.Ar_ptr[k + 1] = P.Ar_ptr[k] + (perm[k] < my ? ar_ptr[perm[k] + 1] - ar_ptr[perm[k]] : 1);
    P.Ar_col.resize(P.Ar_ptr[m]);
    P.Ar_src.resize(P.Ar_ptr[m]);
    for (int k = 0; k < m; ++k) {
      const int b = perm[k];
      if (b >= my) {
        const int d = P.late_cols[(size_t)(b - my)];
        P.Ar_col[P.Ar_ptr[k]] = d;
        P.Ar_src[P.Ar_ptr[k]] = Kp[d];
        continue;
      }
      std::copy(ar_col.begin() + ar_ptr[b], ar_col.begin() + ar_ptr[b + 1], P.Ar_col.begin() + P.Ar_ptr[k]);
      std::copy(ar_src.begin() + ar_ptr[b], ar_src.begin() + ar_ptr[b + 1], P.Ar_src.begin() + P.Ar_ptr[k]);
    }
    huge_resize(P.Kc_y, (size_t)nnz);
    for (int j = 0; j < nx; ++j) {
      P.Kc_y[Kp[j]] = -1;
      for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e) P.Kc_y[e] = iperm[Ki[e] - nx];
    }
  } else {
    P.src.assign(P.Mp[m], -1);
    for (int j = 0; j < N; ++j)
      for (int e = Kp[j]; e < Kp[j + 1]; ++e) {
        const int a = iperm[Ki[e]], b = iperm[j];
        const int i = std::max(a, b), k = std::min(a, b);
        auto first = P.Mi.begin() + P.Mp[k], last = P.Mi.begin() + P.Mp[k + 1];
        auto it = (i == k) ? first : std::lower_bound(first + 1, last, i);
        if (it == last || *it != i) {
          P.error = "internal: K entry missing from M pattern";
          return false;
        }
        P.src[it - P.Mi.begin()] = e;
      }
  }
  tick("value sources (product lists)");
  P.t_symbolic = now_s() - t2;
  P.t_total = now_s() - t0;
  return true;
}


bool build_plan_bounds(int N, const int* Kp, const int* Ki, const double* Kx, const PlanParams& prm, Plan& P) {
  auto plain = [&] {
    const bool ok = build_plan(N, Kp, Ki, Kx, prm, P);
    P.N_ext = N;
    return ok;
  };
  if (N <= 0 || !Kp || !Ki || prm.force_generic) return plain();
  int n = N;
  while (n > 0 && Kp[n] == Kp[n - 1]) --n;
  const int W = N - n;
  if (n == 0 || W == 0) return plain();
  // shape of the leading columns: unit diagonal first, everything else in the (2,1) block, ascending (anything else is
  // build_plan's to classify or refuse)
  std::vector<int> cnt((size_t)W, 0), one((size_t)W, -1);
  for (int j = 0; j < n; ++j) {
    const int a = Kp[j], b = Kp[j + 1];
    if (a >= b || Ki[a] != j || (Kx && Kx[a] != 1.0)) return plain();
    for (int e = a + 1; e < b; ++e) {
      if (Ki[e] < n || Ki[e] >= N || (e > a + 1 && Ki[e] <= Ki[e - 1])) return plain();
      ++cnt[(size_t)(Ki[e] - n)];
      one[(size_t)(Ki[e] - n)] = e;
    }
  }
  // unit rows: one entry (of value one).  Two of them on one variable are a rank-deficient K: not touched here.
  std::vector<int> fixed_by((size_t)n, -1);  // variable -> its unit row
  std::vector<char> is_bound((size_t)W, 0);
  int nb = 0;
  for (int r = 0; r < W; ++r)
    if (cnt[(size_t)r] == 1 && (!Kx || Kx[one[(size_t)r]] == 1.0)) {
      const int j = (int)(std::upper_bound(Kp, Kp + n + 1, one[(size_t)r]) - Kp) - 1;
      if (fixed_by[(size_t)j] >= 0) return plain();
      fixed_by[(size_t)j] = r;
      is_bound[(size_t)r] = 1;
      ++nb;
    }
  if (nb == 0) return plain();
  // K': the other rows renumbered, the fixed variables' columns cut down to their diagonal
  std::vector<int> newrow((size_t)W, -1), row_ext;
  for (int r = 0; r < W; ++r)
    if (!is_bound[(size_t)r]) {
      newrow[(size_t)r] = (int)row_ext.size();
      row_ext.push_back(r);
    }
  const int W2 = (int)row_ext.size(), N2 = n + W2;
  std::vector<int> kp2((size_t)N2 + 1, 0), ki2, ent, bnd_row, bnd_col, cut_ptr{0}, cut_row, cut_ent;
  std::vector<double> kx2;
  ki2.reserve((size_t)Kp[N]);
  ent.reserve((size_t)Kp[N]);
  for (int j = 0; j < n; ++j) {
    ki2.push_back(j);
    ent.push_back(Kp[j]);
    if (fixed_by[(size_t)j] >= 0) {
      bnd_row.push_back(fixed_by[(size_t)j]);
      bnd_col.push_back(j);
      for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e)
        if (!is_bound[(size_t)(Ki[e] - n)]) {
          cut_row.push_back(newrow[(size_t)(Ki[e] - n)]);
          cut_ent.push_back(e);
        }
      cut_ptr.push_back((int)cut_row.size());
    } else {
      for (int e = Kp[j] + 1; e < Kp[j + 1]; ++e) {
        ki2.push_back(n + newrow[(size_t)(Ki[e] - n)]);  // (rows of free columns are never bound rows: those have one entry)
        ent.push_back(e);
      }
    }
    kp2[(size_t)j + 1] = (int)ki2.size();
  }
  for (int j = n; j < N2; ++j) kp2[(size_t)j + 1] = kp2[(size_t)n];
  if (Kx) {
    kx2.resize(ent.size());
    for (size_t e = 0; e < ent.size(); ++e) kx2[e] = Kx[ent[e]];
  }
  const bool ok = build_plan(N2, kp2.data(), ki2.data(), Kx ? kx2.data() : nullptr, prm, P);
  P.N_ext = N;
  P.n_bounds = nb;
  P.bnd_row = std::move(bnd_row);
  P.bnd_col = std::move(bnd_col);
  P.row_ext = std::move(row_ext);
  P.ent_ext = std::move(ent);
  P.cut_ptr = std::move(cut_ptr);
  P.cut_row = std::move(cut_row);
  P.cut_ent = std::move(cut_ent);
  return ok;
}

}  // namespace hipfact
