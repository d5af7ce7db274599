// Undirected graph in CSR form (no self loops, both directions stored).
// Host-side analysis helper for the hipfact KKT backend.
#pragma once
#include <cstdint>
#include <vector>

#include "hostmem.h"

namespace hipfact {

struct Graph {
  int n = 0;
  std::vector<int64_t> ptr;  // n+1
  BigVec<int> adj;           // ptr[n] entries (large: huge pages, no zero fill - hostmem.h)
  int64_t nedges() const { return ptr.empty() ? 0 : ptr[n]; }
};

// Fill-reducing orderings.  perm[k] = vertex eliminated k-th.
//
// amd_order: quotient-graph approximate minimum degree (element absorption,
// approximate external degrees, supervariable detection, mass elimination).
void amd_order(const Graph& g, std::vector<int>& perm);

struct NDParams {
  int leaf_size = 200;        // stop dissecting below this many vertices
  double max_sep_frac = 0.20; // reject separators larger than this fraction of the subgraph
  // ... except in small subgraphs (<= small_k vertices): there the device cost is the NUMBER of tree
  // levels (dependent latency), not the fill - a leaf subgraph wider than one front becomes a chain of
  // fronts, one level each, where one more dissection gives two single-front leaves under one separator
  double small_sep_frac = 0.45;
  int small_k = 2048;
  double balance = 0.15;      // smaller side must hold at least this fraction
  double balance_weight = 0.5; // cost of a cut: separator size x (1 + balance_weight |left - right| / k)
  double depth_tol = 0.4;      // cuts within this of the cheapest compete on the estimated depth of the subtree (0 = off; 0.15 / 0.25 / 0.4 / 0.6: 22 / 21 / 18 / 18 levels on the 2-D grid of the bench, 54 / 48 / 48 / 47 on the 3-D one, at -19 .. -24 % / -15 .. -20 % of the flops of 0)
  // FM-refined edge bisection + minimum vertex cover as a second separator candidate.  Off by
  // default: on the band-like benchmark graphs it shrinks the separators by ~3 % but the resulting
  // trees factor 1-5 % slower (HIPFACT_ND_REFINE=1 to try it on other graph classes).
  bool refine = false;
  double refine_balance = 0.42;  // the FM passes keep each side above this fraction
};

// nd_order: recursive level-structure nested dissection (separators last),
// minimum degree on the leaves.  Exposes elimination-tree parallelism, which is
// what the level-scheduled device factorisation needs.
void nd_order(const Graph& g, const NDParams& p, std::vector<int>& perm);

}  // namespace hipfact
