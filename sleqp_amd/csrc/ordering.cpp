// Fill-reducing orderings for the hipfact KKT backend (host side, cached per
// sparsity pattern).
//
// The reference delegates ordering to its third-party solvers (MA57: AMD/MC47,
// fact_ma57.c:761-763; MA86/MA97: mc68_order, fact_ma86.c:200, fact_ma97.c:280;
// CHOLMOD default ordering, fact_cholmod.c:133).  This backend owns it:
//   * amd_order: quotient-graph approximate minimum degree.
//   * nd_order : recursive level-structure nested dissection with minimum
//                degree leaves.  Nested dissection is what makes the
//                elimination tree short and bushy, i.e. what gives the
//                level-scheduled device factorisation / solves their
//                parallelism.
#include "graph.h"

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <queue>
#include <thread>
#include <vector>

namespace hipfact {

namespace {

enum : uint8_t { ST_VAR = 0, ST_ELEM = 1, ST_DEAD = 2, ST_ABSORBED = 3 };

struct DegLists {
  std::vector<int> head, next, prev;
  int mindeg;
  explicit DegLists(int n) : head(n + 1, -1), next(n, -1), prev(n, -1), mindeg(n) {}
  void insert(int i, int d) {
    next[i] = head[d];
    prev[i] = -1;
    if (head[d] != -1) prev[head[d]] = i;
    head[d] = i;
    if (d < mindeg) mindeg = d;
  }
  void remove(int i, int d) {
    if (prev[i] != -1)
      next[prev[i]] = next[i];
    else
      head[d] = next[i];
    if (next[i] != -1) prev[next[i]] = prev[i];
    next[i] = prev[i] = -1;
  }
};

}  // namespace

void amd_order(const Graph& g, std::vector<int>& perm) {
  const int n = g.n;
  perm.clear();
  perm.reserve(n);
  if (n == 0) return;

  std::vector<std::vector<int>> avar(n), aelm(n), evars(n);
  std::vector<uint8_t> status(n, ST_VAR);
  std::vector<int> nv(n, 1), degree(n), esize(n, 0);
  std::vector<int> mark(n, 0), wflag(n, 0), w(n, 0);
  // supervariable member chains (for the final ordering)
  std::vector<int> chain_next(n, -1), chain_tail(n);
  std::iota(chain_tail.begin(), chain_tail.end(), 0);

  DegLists dl(n);
  for (int i = 0; i < n; ++i) {
    avar[i].assign(g.adj.begin() + g.ptr[i], g.adj.begin() + g.ptr[i + 1]);
    degree[i] = (int)avar[i].size();
    dl.insert(i, degree[i]);
  }

  int tag = 0, wtag = 0, nel = 0;
  std::vector<int> Lp;
  std::vector<std::pair<uint32_t, int>> hashes;

  while (nel < n) {
    // ---- pick pivot of minimum approximate degree
    while (dl.head[dl.mindeg] == -1) ++dl.mindeg;
    const int p = dl.head[dl.mindeg];
    dl.remove(p, degree[p]);

    // ---- form the new element L_p
    ++tag;
    mark[p] = tag;
    Lp.clear();
    for (int v : avar[p])
      if (status[v] == ST_VAR && nv[v] > 0 && mark[v] != tag) {
        mark[v] = tag;
        Lp.push_back(v);
      }
    for (int e : aelm[p]) {
      if (status[e] != ST_ELEM) continue;
      for (int v : evars[e])
        if (status[v] == ST_VAR && nv[v] > 0 && mark[v] != tag) {
          mark[v] = tag;
          Lp.push_back(v);
        }
      status[e] = ST_DEAD;  // absorbed into p
      std::vector<int>().swap(evars[e]);
    }
    std::vector<int>().swap(avar[p]);
    std::vector<int>().swap(aelm[p]);
    status[p] = ST_ELEM;
    nel += nv[p];
    const int nleft = n - nel;
    int degme = 0;
    for (int v : Lp) {
      degme += nv[v];
      dl.remove(v, degree[v]);
    }
    esize[p] = degme;

    // ---- prune adjacency of the members, attach the new element
    for (int i : Lp) {
      auto& ae = aelm[i];
      size_t k = 0;
      for (int e : ae)
        if (status[e] == ST_ELEM && e != p) ae[k++] = e;
      ae.resize(k);
      auto& av = avar[i];
      k = 0;
      for (int v : av)
        if (status[v] == ST_VAR && nv[v] > 0 && mark[v] != tag) av[k++] = v;
      av.resize(k);
    }

    // ---- |L_e \ L_p| for every element adjacent to a member
    ++wtag;
    for (int i : Lp)
      for (int e : aelm[i]) {
        if (wflag[e] != wtag) {
          wflag[e] = wtag;
          w[e] = esize[e];
        }
        w[e] -= nv[i];
      }

    // ---- approximate degrees, aggressive absorption, hashing
    hashes.clear();
    for (int i : Lp) {
      auto& ae = aelm[i];
      size_t k = 0;
      long long deg = 0;
      uint32_t h = 0;
      for (int e : ae) {
        if (status[e] != ST_ELEM) continue;
        if (w[e] <= 0) {  // L_e subset of L_p: absorb e into p
          status[e] = ST_DEAD;
          std::vector<int>().swap(evars[e]);
          continue;
        }
        deg += w[e];
        h += (uint32_t)e;
        ae[k++] = e;
      }
      ae.resize(k);
      ae.push_back(p);
      h += (uint32_t)p;
      for (int v : avar[i]) {
        deg += nv[v];
        h += (uint32_t)v;
      }
      if (deg > n) deg = n;
      degree[i] = std::min<long long>(degree[i], deg);  // size of L_p added below
      hashes.emplace_back(h, i);
    }

    // ---- supervariable detection among the members of L_p
    std::sort(hashes.begin(), hashes.end());
    for (size_t a = 0; a < hashes.size();) {
      size_t b = a;
      while (b < hashes.size() && hashes[b].first == hashes[a].first) ++b;
      for (size_t s = a; s < b; ++s) {
        const int i = hashes[s].second;
        if (nv[i] == 0) continue;
        ++tag;
        for (int e : aelm[i]) mark[e] = tag;
        // variables and elements share the id space [0,n): a vertex is either a
        // live variable or an element, never both, so one marker array suffices
        for (int v : avar[i]) mark[v] = tag;
        for (size_t t = s + 1; t < b; ++t) {
          const int j = hashes[t].second;
          if (nv[j] == 0) continue;
          if (aelm[j].size() != aelm[i].size() || avar[j].size() != avar[i].size()) continue;
          bool same = true;
          for (int e : aelm[j])
            if (mark[e] != tag) {
              same = false;
              break;
            }
          if (same)
            for (int v : avar[j])
              if (mark[v] != tag) {
                same = false;
                break;
              }
          if (!same) continue;
          // j is indistinguishable from i: merge
          nv[i] += nv[j];
          nv[j] = 0;
          status[j] = ST_ABSORBED;
          chain_next[chain_tail[i]] = j;
          chain_tail[i] = chain_tail[j];
          std::vector<int>().swap(avar[j]);
          std::vector<int>().swap(aelm[j]);
        }
      }
      a = b;
    }

    // ---- finalise degrees, rebuild element member list without absorbed vars
    {
      size_t k = 0;
      for (int i : Lp) {
        if (nv[i] == 0) continue;
        long long d = (long long)degree[i] + degme - nv[i];
        d = std::min<long long>(d, nleft - nv[i]);
        if (d < 0) d = 0;
        degree[i] = (int)d;
        dl.insert(i, degree[i]);
        Lp[k++] = i;
      }
      Lp.resize(k);
      evars[p] = Lp;
    }

    // ---- emit p and the variables merged into it
    for (int v = p; v != -1; v = chain_next[v]) perm.push_back(v);
  }
  assert((int)perm.size() == n);
}

// ---------------------------------------------------------------------------
// nested dissection
// ---------------------------------------------------------------------------
namespace {

struct NDState {
  const Graph& g;
  NDParams prm;
  std::vector<int> inset;  // stamp: vertex belongs to the current subset
  std::vector<int> level;  // BFS level / scratch
  std::vector<int> local;  // global -> local index for leaf subgraphs
  std::atomic<int> stamp{0};
  std::vector<int>& perm;  // pre-sized: every call of rec() owns the slice [off, off + |verts|)
  // Leaf subgraphs are independent: the dissection only records them (vertex
  // list + the slice of the permutation they own, marked in leaf_id[]) and they
  // are ordered by minimum degree in parallel afterwards.
  struct LeafTask {
    std::vector<int> verts;
    int off;
  };
  std::vector<LeafTask> leaves;
  std::vector<int> leaf_id;
  NDState(const Graph& g_, const NDParams& p, std::vector<int>& out)
      : g(g_), prm(p), inset(g_.n, -1), level(g_.n, -1), local(g_.n, -1), perm(out), leaf_id(g_.n, -1) {
    if (p.refine) {
      side.assign(g.n, 0);
      ext.assign(g.n, 0);
      deg.assign(g.n, 0);
      lockst.assign(g.n, 0);
      mate.assign(g.n, -1);
      hk_dist.assign(g.n, 0);
      in_cover.assign(g.n, 0);
    }
  }

  // The two halves of a dissection are independent (disjoint vertex sets, disjoint slices of the
  // permutation, unique stamps): large ones run on their own threads.
  std::mutex leaves_mutex;
  std::atomic<int> threads_running{1};
  int max_threads = 1;
  const bool nd_debug = getenv("HIPFACT_ND_DEBUG") != nullptr;

  void leaf(const std::vector<int>& verts, int off) {
    int id;
    {
      std::lock_guard<std::mutex> lock(leaves_mutex);
      id = (int)leaves.size();
      if (verts.size() > 2) leaves.push_back({verts, off});
      else leaves.push_back({std::vector<int>(), off});
    }
    for (size_t t = 0; t < verts.size(); ++t) {
      perm[off + t] = verts[t];  // placeholder order, replaced by run_leaves()
      leaf_id[verts[t]] = id;
    }
  }

  void order_leaf(const LeafTask& task, int id) {
    const std::vector<int>& verts = task.verts;
    const int k = (int)verts.size();
    if (k <= 2) return;
    for (int t = 0; t < k; ++t) local[verts[t]] = t;
    Graph sub;
    sub.n = k;
    sub.ptr.assign(k + 1, 0);
    for (int t = 0; t < k; ++t) {
      const int v = verts[t];
      int64_t c = 0;
      for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q)
        if (leaf_id[g.adj[q]] == id) ++c;
      sub.ptr[t + 1] = sub.ptr[t] + c;
    }
    sub.adj.resize(sub.ptr[k]);
    for (int t = 0; t < k; ++t) {
      const int v = verts[t];
      int64_t o = sub.ptr[t];
      for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q)
        if (leaf_id[g.adj[q]] == id) sub.adj[o++] = local[g.adj[q]];
    }
    std::vector<int> lp;
    amd_order(sub, lp);
    for (int t = 0; t < k; ++t) perm[task.off + t] = verts[lp[t]];
  }

  void run_leaves() {
    const int nl = (int)leaves.size();
    const int nt = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
    std::atomic<int> next{0};
    auto worker = [&]() {
      for (int i = next++; i < nl; i = next++) order_leaf(leaves[i], i);
    };
    if (nt <= 1 || nl <= 1) {
      worker();
      return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < std::min(nt, nl); ++t) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
  }

  // BFS inside the stamped subset from root; returns vertices in BFS order and
  // fills level[]; lev_ptr delimits the level sets.
  void bfs(int root, int id, std::vector<int>& order, std::vector<int>& lev_ptr) {
    order.clear();
    lev_ptr.clear();
    order.push_back(root);
    level[root] = 0;
    inset[root] = -id;  // visited marker (negated stamp)
    lev_ptr.push_back(0);
    size_t head = 0;
    int cur = 0;
    while (head < order.size()) {
      const int v = order[head];
      if (level[v] != cur) {
        cur = level[v];
        lev_ptr.push_back((int)head);
      }
      ++head;
      for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
        const int u = g.adj[q];
        if (inset[u] == id) {
          inset[u] = -id;
          level[u] = cur + 1;
          order.push_back(u);
        }
      }
    }
    lev_ptr.push_back((int)order.size());
    for (int v : order) inset[v] = id;  // restore
  }

  // ---- refined bisection --------------------------------------------------
  // side / ext / deg are only valid for the vertices of the current subset.
  std::vector<int8_t> side;
  std::vector<int> ext, deg, lockst, mate, hk_dist;

  void move_vertex(int v, int id) {
    const int a = side[v];
    side[v] = (int8_t)(1 - a);
    ext[v] = deg[v] - ext[v];
    for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
      const int u = g.adj[q];
      if (inset[u] != id) continue;
      ext[u] += (side[u] == a) ? 1 : -1;
    }
  }

  // One FM pass on the edge cut; returns the improvement (>= 0).
  long long fm_pass(const std::vector<int>& verts, int id, int (&cnt)[2], int min_side) {
    struct Ent {
      int gain, v;
      bool operator<(const Ent& o) const { return gain < o.gain || (gain == o.gain && v > o.v); }
    };
    std::priority_queue<Ent> heap[2];
    const int ls = ++stamp;  // unique across threads; lockst[] is only compared for equality
    for (int v : verts)
      if (ext[v] > 0) heap[side[v]].push({2 * ext[v] - deg[v], v});
    std::vector<int> moved;
    long long cur = 0, best = 0;
    size_t best_len = 0;
    const size_t limit = std::max<size_t>(64, verts.size() / 100);
    auto top = [&](int s) -> bool {  // drop stale / locked entries
      auto& h = heap[s];
      while (!h.empty()) {
        const Ent e = h.top();
        if (lockst[e.v] == ls || side[e.v] != s || e.gain != 2 * ext[e.v] - deg[e.v])
          h.pop();
        else
          return true;
      }
      return false;
    };
    while (moved.size() - best_len < limit) {
      const bool ok0 = cnt[0] - 1 >= min_side && top(0);
      const bool ok1 = cnt[1] - 1 >= min_side && top(1);
      if (!ok0 && !ok1) break;
      int s;
      if (ok0 && ok1) {
        const int g0 = heap[0].top().gain, g1 = heap[1].top().gain;
        s = g0 != g1 ? (g0 > g1 ? 0 : 1) : (cnt[0] >= cnt[1] ? 0 : 1);
      } else {
        s = ok0 ? 0 : 1;
      }
      const Ent e = heap[s].top();
      heap[s].pop();
      const int v = e.v;
      lockst[v] = ls;
      cur += e.gain;
      move_vertex(v, id);
      --cnt[s];
      ++cnt[1 - s];
      moved.push_back(v);
      for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
        const int u = g.adj[q];
        if (inset[u] != id || lockst[u] == ls) continue;
        if (ext[u] > 0) heap[side[u]].push({2 * ext[u] - deg[u], u});
      }
      if (cur > best) {
        best = cur;
        best_len = moved.size();
      }
    }
    for (size_t t = moved.size(); t > best_len; --t) {
      const int v = moved[t - 1];
      const int s = side[v];
      move_vertex(v, id);
      --cnt[s];
      ++cnt[1 - s];
    }
    return best;
  }

  // Hopcroft-Karp on the bipartite graph of the cut edges (X: side 0, Y: side 1), then the
  // Koenig cover: the smallest vertex set meeting every cut edge, i.e. a vertex separator.
  void min_cover(const std::vector<int>& X, int id, std::vector<char>& in_cover) {
    // mate[] / hk_dist[] are indexed by global vertex; only boundary vertices are touched
    const int INF = 1 << 30;
    std::vector<int> queue;
    auto cross = [&](int v, int u) { return inset[u] == id && side[u] != side[v]; };
    for (;;) {
      queue.clear();
      for (int x : X) {
        if (mate[x] < 0) {
          hk_dist[x] = 0;
          queue.push_back(x);
        } else {
          hk_dist[x] = INF;
        }
      }
      bool found = false;
      for (size_t h = 0; h < queue.size(); ++h) {
        const int x = queue[h];
        for (int64_t q = g.ptr[x]; q < g.ptr[x + 1]; ++q) {
          const int y = g.adj[q];
          if (!cross(x, y)) continue;
          const int x2 = mate[y];
          if (x2 < 0) {
            found = true;
          } else if (hk_dist[x2] == INF) {
            hk_dist[x2] = hk_dist[x] + 1;
            queue.push_back(x2);
          }
        }
      }
      if (!found) break;
      // layered DFS (iterative) from every free x
      struct Fr {
        int x;
        int64_t q;
      };
      std::vector<Fr> st;
      for (int x0 : X) {
        if (mate[x0] >= 0) continue;
        st.clear();
        st.push_back({x0, g.ptr[x0]});
        while (!st.empty()) {
          Fr& f = st.back();
          const int x = f.x;
          bool advanced = false;
          while (f.q < g.ptr[x + 1]) {
            const int y = g.adj[f.q++];
            if (!cross(x, y)) continue;
            const int x2 = mate[y];
            if (x2 < 0) {
              // augment along the stack
              int yy = y;
              for (size_t t = st.size(); t > 0; --t) {
                const int xx = st[t - 1].x;
                const int prev = mate[xx];
                mate[xx] = yy;
                mate[yy] = xx;
                yy = prev;
              }
              st.clear();
              advanced = true;
              break;
            }
            if (hk_dist[x2] == hk_dist[x] + 1) {
              st.push_back({x2, g.ptr[x2]});
              advanced = true;
              break;
            }
          }
          if (!advanced) {
            hk_dist[x] = INF;  // dead end
            st.pop_back();
          }
        }
      }
    }
    // Koenig: Z = reachable from free X by alternating paths; cover = (X \ Z) + (Y & Z)
    queue.clear();
    for (int x : X) {
      hk_dist[x] = 0;
      if (mate[x] < 0) {
        hk_dist[x] = 1;
        queue.push_back(x);
      }
    }
    for (size_t h = 0; h < queue.size(); ++h) {
      const int x = queue[h];
      for (int64_t q = g.ptr[x]; q < g.ptr[x + 1]; ++q) {
        const int y = g.adj[q];
        if (!cross(x, y) || in_cover[y]) continue;
        in_cover[y] = 1;  // y in Z
        const int x2 = mate[y];
        if (x2 >= 0 && !hk_dist[x2]) {
          hk_dist[x2] = 1;
          queue.push_back(x2);
        }
      }
    }
    for (int x : X)
      if (!hk_dist[x]) in_cover[x] = 1;
  }

  std::vector<char> in_cover;

  bool refined_cut(const std::vector<int>& order, int id, std::vector<int>& left, std::vector<int>& right,
                   std::vector<int>& sep) {
    const int k = (int)order.size();
    int cnt[2] = {k / 2, k - k / 2};
    for (int t = 0; t < k; ++t) side[order[t]] = (int8_t)(t >= cnt[0]);
    for (int v : order) {
      int d = 0, e = 0;
      for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
        const int u = g.adj[q];
        if (inset[u] != id) continue;
        ++d;
        e += side[u] != side[v];
      }
      deg[v] = d;
      ext[v] = e;
    }
    const int min_side = std::max(1, (int)(prm.refine_balance * k));
    for (int pass = 0; pass < 12; ++pass)
      if (fm_pass(order, id, cnt, min_side) <= 0) break;
    // boundary and cover
    std::vector<int> X, B;
    for (int v : order)
      if (ext[v] > 0) {
        B.push_back(v);
        if (side[v] == 0) X.push_back(v);
      }
    if (X.empty()) return false;
    for (int v : B) {
      mate[v] = -1;
      in_cover[v] = 0;
    }
    min_cover(X, id, in_cover);
    for (int v : order) {
      if (ext[v] > 0 && in_cover[v])
        sep.push_back(v);
      else
        (side[v] == 0 ? left : right).push_back(v);
    }
    for (int v : B) {
      mate[v] = -1;
      in_cover[v] = 0;
    }
    const int small = (int)std::min(left.size(), right.size());
    return small >= prm.balance * k;
  }

  void rec(std::vector<int> verts, int off, int depth = 0) {
    const int k = (int)verts.size();
    if (k <= prm.leaf_size) {
      leaf(verts, off);
      return;
    }
    const int id = ++stamp;
    for (int v : verts) inset[v] = id;
    const auto t_rec0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rec0).count(); };

    std::vector<int> order, lev_ptr;
    // connected components: only looked for when the first sweep below does not reach every vertex
    auto split_components = [&]() -> bool {
      std::vector<std::vector<int>> comps;
      const int cid = ++stamp;  // component-visited stamp
      std::vector<int> queue;
      for (int s : verts) {
        if (inset[s] != id) continue;
        queue.clear();
        queue.push_back(s);
        inset[s] = cid;
        for (size_t h = 0; h < queue.size(); ++h) {
          const int v = queue[h];
          for (int64_t q = g.ptr[v]; q < g.ptr[v + 1]; ++q) {
            const int u = g.adj[q];
            if (inset[u] == id) {
              inset[u] = cid;
              queue.push_back(u);
            }
          }
        }
        if ((int)queue.size() == k) break;  // single component
        comps.push_back(queue);
      }
      if (!comps.empty()) {
        std::vector<int>().swap(verts);
        int o = off;
        for (auto& c : comps) {
          const int kc = (int)c.size();
          rec(std::move(c), o, depth);
          o += kc;
        }
        return true;
      }
      for (int v : verts) inset[v] = id;
      return false;
    };

    // ---- pseudo-peripheral root
    int root = verts[0];
    {
      int64_t best = INT64_MAX;
      for (int v : verts) {
        const int64_t d = g.ptr[v + 1] - g.ptr[v];
        if (d < best) {
          best = d;
          root = v;
        }
      }
    }
    int ecc = -1;
    bool fresh = false;  // order / lev_ptr belong to the current root
    // (usually converges after two or three sweeps; every sweep of a large subgraph is serial time in front of the
    // parallel part of the dissection)
    for (int it = 0; it < (k > 16384 ? 3 : 6); ++it) {
      bfs(root, id, order, lev_ptr);
      fresh = true;
      if (it == 0 && (int)order.size() < k) {
        if (split_components()) return;
      }
      const int nlev = (int)lev_ptr.size() - 1;
      if (nlev - 1 <= ecc) break;
      ecc = nlev - 1;
      // min-degree vertex of the last level
      int cand = order[lev_ptr[nlev - 1]];
      int64_t best = INT64_MAX;
      for (int t = lev_ptr[nlev - 1]; t < lev_ptr[nlev]; ++t) {
        const int v = order[t];
        const int64_t d = g.ptr[v + 1] - g.ptr[v];
        if (d < best) {
          best = d;
          cand = v;
        }
      }
      if (cand == root) break;
      root = cand;
      fresh = false;
    }
    if (!fresh) bfs(root, id, order, lev_ptr);
    const double t_bfs = since();
    const int nlev = (int)lev_ptr.size() - 1;
    if (nlev < 3) {
      leaf(verts, off);
      return;
    }

    // ---- candidate 1: smallest level set among balanced cuts, trimmed
    std::vector<int> left, right, sep;
    {
      int best_l = -1;
      double best_cost = 1e300;
      for (int l = 1; l + 1 < nlev; ++l) {
        const int before = lev_ptr[l];
        const int sz = lev_ptr[l + 1] - lev_ptr[l];
        const int after = k - before - sz;
        const int small = std::min(before, after);
        if (small < prm.balance * k) continue;
        const double cost = (double)sz * (1.0 + prm.balance_weight * std::abs(before - after) / (double)k);
        if (cost < best_cost) {
          best_cost = cost;
          best_l = l;
        }
      }
      // The device pays for every LEVEL of the tree (a dependent pivot block, ~25 us) before it pays for fill: among
      // the cuts within depth_tol of the cheapest, take the one under which the deeper side needs the fewest further
      // dissections - estimated with separators of this cut's size and leaves of one front each: a subgraph of k
      // vertices needs d more levels when k <= cap(d), cap(0) = leaf_size, cap(d) = 2 cap(d - 1) + sz.  (A band of
      // 5e4 rows with 84-row separators fits 256 single-front leaves under 8 levels of separators - if no cut
      // leaves more than cap(d - 1) on one side; the cheapest cut alone gave 11 levels, two of them from sides that
      // missed their capacity by a few per cent.)
      if (best_l >= 0 && prm.depth_tol > 0.0) {
        auto need = [&](int kk, int sz) {
          int d = 0;
          for (long long cap = prm.leaf_size; cap < kk && d < 40; ++d) cap = 2 * cap + sz;
          return d;
        };
        int best_d = 1 << 30;
        double best_c2 = 1e300;
        int pick = best_l;
        for (int l = 1; l + 1 < nlev; ++l) {
          const int before = lev_ptr[l];
          const int sz = lev_ptr[l + 1] - lev_ptr[l];
          const int after = k - before - sz;
          if (std::min(before, after) < prm.balance * k) continue;
          const double cost = (double)sz * (1.0 + prm.balance_weight * std::abs(before - after) / (double)k);
          if (cost > (1.0 + prm.depth_tol) * best_cost) continue;
          const int d = std::max(need(before, sz), need(after, sz));
          if (d < best_d || (d == best_d && cost < best_c2)) {
            best_d = d;
            best_c2 = cost;
            pick = l;
          }
        }
        best_l = pick;
      }
      if (best_l >= 0) {
        left.assign(order.begin(), order.begin() + lev_ptr[best_l]);
        right.assign(order.begin() + lev_ptr[best_l + 1], order.end());
        // trim: separator vertices without a neighbour in the next level go left
        for (int t = lev_ptr[best_l]; t < lev_ptr[best_l + 1]; ++t) {
          const int v = order[t];
          bool touches = false;
          for (int64_t q = g.ptr[v]; q < g.ptr[v + 1] && !touches; ++q) {
            const int u = g.adj[q];
            touches = (inset[u] == id && level[u] == best_l + 1);
          }
          if (touches)
            sep.push_back(v);
          else
            left.push_back(v);
        }
      }
    }
    // ---- candidate 2: edge bisection of the BFS order refined by FM, then a minimum vertex
    // cover of the cut edges.  Level sets of graphs with long-range edges (band-like M = A A^T)
    // are smeared over about twice the width of the thinnest separator.
    // (only where the level structure found a cut: otherwise the subgraph is small or has no
    // thin separator, and another tree level costs more than it saves)
    if (prm.refine && !sep.empty()) {
      std::vector<int> l2, r2, s2;
      const bool okr = refined_cut(order, id, l2, r2, s2);
      if (getenv("HIPFACT_ND_DEBUG"))
        fprintf(stderr, "nd k=%d levelset sep %zu (l %zu r %zu) refined ok %d sep %zu (l %zu r %zu)\n", k, sep.size(),
                left.size(), right.size(), (int)okr, s2.size(), l2.size(), r2.size());
      if (okr && s2.size() < sep.size()) {
        left.swap(l2);
        right.swap(r2);
        sep.swap(s2);
      }
    }
    if (sep.empty() || left.empty() || right.empty() || (double)sep.size() > (k <= prm.small_k ? std::max(prm.max_sep_frac, prm.small_sep_frac) : prm.max_sep_frac) * k) {
      leaf(verts, off);
      return;
    }
    std::vector<int>().swap(verts);
    std::vector<int>().swap(order);
    const int nl = (int)left.size(), nr = (int)right.size();
    if (nd_debug && depth <= 3)
      fprintf(stderr, "nd depth %d k %d sep %zu left %d right %d nlev %d: sweeps %.2f ms, cut %.2f ms\n", depth, k, sep.size(), nl, nr,
              nlev, t_bfs, since() - t_bfs);
    for (size_t t = 0; t < sep.size(); ++t) perm[off + nl + nr + t] = sep[t];
    if (std::min(nl, nr) >= 2048 && threads_running.load() < max_threads) {
      ++threads_running;
      std::thread other([this, &left, off, depth] { rec(std::move(left), off, depth + 1); });
      rec(std::move(right), off + nl, depth + 1);
      other.join();
      --threads_running;
    } else {
      rec(std::move(left), off, depth + 1);
      rec(std::move(right), off + nl, depth + 1);
    }
  }
};

}  // namespace

void nd_order(const Graph& g, const NDParams& p, std::vector<int>& perm) {
  perm.assign(g.n, -1);
  NDState st(g, p, perm);
  st.max_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
  std::vector<int> all(g.n);
  std::iota(all.begin(), all.end(), 0);
  const auto t0 = std::chrono::steady_clock::now();
  st.rec(std::move(all), 0);
  const auto t1 = std::chrono::steady_clock::now();
  st.run_leaves();
  if (st.nd_debug)
    fprintf(stderr, "nd: dissection %.2f ms, %zu leaves ordered in %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
            st.leaves.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  assert((int)perm.size() == g.n);
}

}  // namespace hipfact
