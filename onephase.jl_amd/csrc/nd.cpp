// Level-structure nested dissection for path-like graphs (banded KKT systems: BASELINE config 2, the CUTEst CHAIN family).
//
// Why: a minimum-degree ordering of a banded matrix eliminates from the ends inwards -- zero fill, but the elimination tree
// is a path: n dependent pivots, nothing for a GPU to do in parallel (S-C2 at N_h = 20000: 71 ms for 2.4 MFLOP, 20x slower
// than one CPU core).  The reference inherits CHOLMOD's AMD (julia.jl:34,52) because it runs on one core; the pivot order is
// free for a static-pivot LDL^T of a quasi-definite matrix, so the analysis may pick one that exposes parallelism.
//
// Algorithm (George's automatic nested dissection): breadth-first level structure from a pseudo-peripheral node; the level
// that halves the node count is a vertex separator; recurse on both sides; order = [left, right, separator].  The tree is
// balanced (depth log2(n / leaf)) and on a band of width w every separator has about w nodes: fill O(n w log n).
// Deterministic: ties are broken by node number.
#include "symbolic.h"

#include <algorithm>
#include <cmath>

namespace okkt {

namespace {

struct LevelNd {
  int n;
  const std::vector<int64_t>& gp;
  const std::vector<int>& gi;
  int leaf;
  std::vector<int> tag;      // region id of every node (-1: already ordered)
  std::vector<int> dist;     // BFS scratch
  std::vector<int> queue;
  std::vector<int>& order;
  int next_tag = 1;

  LevelNd(int n_, const std::vector<int64_t>& gp_, const std::vector<int>& gi_, int leaf_, std::vector<int>& order_)
      : n(n_), gp(gp_), gi(gi_), leaf(leaf_), tag(n_, 0), dist(n_, -1), order(order_) { queue.reserve(n_); }

  // BFS inside region t from root; fills queue (visit order) and dist; returns the number of levels
  int bfs(int root, int t) {
    queue.clear();
    queue.push_back(root);
    dist[root] = 0;
    int nlev = 1;
    for (size_t h = 0; h < queue.size(); ++h) {
      const int v = queue[h];
      for (int64_t p = gp[v]; p < gp[v + 1]; ++p) {
        const int u = gi[p];
        if (tag[u] == t && dist[u] < 0) { dist[u] = dist[v] + 1; nlev = dist[u] + 1; queue.push_back(u); }
      }
    }
    return nlev;
  }
  void clear_dist() { for (int v : queue) dist[v] = -1; }
  int degree_in(int v, int t) const {
    int d = 0;
    for (int64_t p = gp[v]; p < gp[v + 1]; ++p) d += tag[gi[p]] == t;
    return d;
  }

  void dissect(const std::vector<int>& V, int t) {
    // connected components first: they are independent subtrees, no separator needed
    for (size_t cursor = 0; cursor < V.size(); ++cursor) {
      if (tag[V[cursor]] != t) continue;     // reached from an earlier root
      int root = V[cursor];
      int nlev = bfs(root, t);
      // pseudo-peripheral root: restart from a minimum-degree node of the last level while the structure gets deeper
      for (int it = 0; it < 4; ++it) {
        int best = -1, bestd = 0;
        for (size_t q = queue.size(); q-- > 0;) {
          const int v = queue[q];
          if (dist[v] != nlev - 1) break;
          const int d = degree_in(v, t);
          if (best < 0 || d < bestd || (d == bestd && v < best)) { best = v; bestd = d; }
        }
        if (best < 0 || best == root) break;
        clear_dist();
        const int nlev2 = bfs(best, t);
        root = best;
        if (nlev2 <= nlev) { nlev = nlev2; break; }
        nlev = nlev2;
      }
      std::vector<int> comp(queue);          // this component in BFS order
      const int nc = (int)comp.size();
      if (nc <= leaf || nlev < 3) {
        // leaf: eliminate in breadth-first order (a band elimination inside the piece)
        for (int v : comp) { order.push_back(v); tag[v] = -1; }
        clear_dist();
      } else {
        // separator = the level at which the cumulative count passes half of the component
        std::vector<int> lev_cnt(nlev, 0);
        for (int v : comp) ++lev_cnt[dist[v]];
        int ls = 1, acc = lev_cnt[0];
        while (ls < nlev - 2 && acc + lev_cnt[ls] < nc / 2) acc += lev_cnt[ls++];
        // a narrower level nearby is a cheaper separator as long as the halves stay within 2 : 1
        {
          int below = 0, best = ls;
          for (int l = 1; l <= nlev - 2; ++l) {
            below += lev_cnt[l - 1];
            const int above = nc - below - lev_cnt[l];
            if (below * 2 >= above && above * 2 >= below && lev_cnt[l] < lev_cnt[best]) best = l;
          }
          ls = best;
        }
        std::vector<int> A, B, Sep;
        const int ta = next_tag++, tb = next_tag++;
        for (int v : comp) {
          const int d = dist[v];
          if (d < ls) { A.push_back(v); tag[v] = ta; }
          else if (d > ls) { B.push_back(v); tag[v] = tb; }
          else Sep.push_back(v);
        }
        clear_dist();
        for (int v : Sep) tag[v] = -1;   // out of both halves before they are dissected
        dissect(A, ta);
        dissect(B, tb);
        std::sort(Sep.begin(), Sep.end());
        for (int v : Sep) order.push_back(v);
      }
    }
  }
};

}  // namespace

void level_nd_order(int n, const std::vector<int64_t>& gp, const std::vector<int>& gi, int leaf, std::vector<int>& order) {
  order.clear();
  order.reserve(n);
  LevelNd nd(n, gp, gi, std::max(leaf, 4), order);
  // dense rows (a constraint over all variables, e.g. the length constraint of the hanging chain) would make every level
  // structure two levels deep: they leave the graph first and are eliminated last, as in the minimum-degree code
  // Threshold: 10 sqrt(n) as in the minimum-degree code, but not more than 8 x the average degree -- on a small system 10 sqrt(n)
  // is most of the graph (hanging chain, N_h = 300: the length constraint has 301 entries, 10 sqrt(1208) = 348, and the level
  // structure stayed two levels deep; at the CUTEst size N_h = 400 it passed by one entry).  Being generous is safe here: the caller
  // keeps this ordering only if its elimination tree is three times shallower than minimum degree's at bounded extra flops.
  const double avg_deg = n > 0 ? (double)gp[n] / (double)n : 0.0;
  const double dense = std::max(16.0, std::min(10.0 * std::sqrt((double)n), 8.0 * avg_deg));
  std::vector<int> all, last;
  for (int i = 0; i < n; ++i) {
    if ((double)(gp[i + 1] - gp[i]) > dense) { last.push_back(i); nd.tag[i] = -1; }
    else all.push_back(i);
  }
  nd.dissect(all, 0);
  for (int v : last) order.push_back(v);
}

}  // namespace okkt
