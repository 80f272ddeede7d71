// Fill-reducing ordering for the KKT factorisation: an approximate-minimum-degree
// ordering on the quotient graph (algorithm of Amestoy, Davis & Duff, SIMAX 1996),
// written from the paper's description for this project.
//
// Role in the path: the reference never passes a permutation to CHOLMOD
// (/root/reference/src/linear_system_solvers/julia.jl:34,52), so CHOLMOD's default
// AMD ordering is applied on every ls_factor! call.  Here the ordering is computed
// once per sparsity pattern on the host and reused by every numeric refactorisation.
//
// Input : symmetric pattern without diagonal, both triangles, CSR/CSC (n, ap, ai).
// Output: order[k] = the variable eliminated k-th.
#include "symbolic.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace okkt {

namespace {

enum : uint8_t { ST_VAR = 0, ST_ELEM = 1, ST_DEAD_ELEM = 2, ST_ABSORBED = 3, ST_DENSE = 4 };

struct DegreeLists {
  std::vector<int> head, next, prev;
  int mindeg;
  explicit DegreeLists(int n) : head(n + 1, -1), next(n, -1), prev(n, -1), mindeg(0) {}
  void insert(int i, int d) {
    next[i] = head[d];
    prev[i] = -1;
    if (head[d] >= 0) prev[head[d]] = i;
    head[d] = i;
    if (d < mindeg) mindeg = d;
  }
  void remove(int i, int d) {
    if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d] = next[i];
    if (next[i] >= 0) prev[next[i]] = prev[i];
  }
};

}  // namespace

void amd_order(int n, const std::vector<int64_t>& ap, const std::vector<int>& ai,
               std::vector<int>& order) {
  order.clear();
  order.reserve(n);
  if (n == 0) return;

  std::vector<std::vector<int>> adjv(n), adje(n), elvars(n);
  std::vector<int> nv(n, 1), degree(n, 0), elemdeg(n, 0);
  std::vector<uint8_t> state(n, ST_VAR);
  std::vector<std::vector<int>> absorbed(n);  // variables ordered together with i

  // dense rows are taken out and ordered last (they would join every element)
  const int dense_thr = std::max(16, (int)(10.0 * std::sqrt((double)n)));
  std::vector<int> dense_nodes;
  for (int i = 0; i < n; ++i) {
    int64_t d = ap[i + 1] - ap[i];
    if (d > dense_thr) { state[i] = ST_DENSE; dense_nodes.push_back(i); }
  }
  int nleft = n - (int)dense_nodes.size();
  for (int i = 0; i < n; ++i) {
    if (state[i] == ST_DENSE) continue;
    auto& a = adjv[i];
    a.reserve(ap[i + 1] - ap[i]);
    for (int64_t p = ap[i]; p < ap[i + 1]; ++p) {
      int j = ai[p];
      if (j != i && state[j] != ST_DENSE) a.push_back(j);
    }
    degree[i] = (int)a.size();
  }

  DegreeLists dl(n);
  dl.mindeg = n;
  for (int i = n - 1; i >= 0; --i)
    if (state[i] == ST_VAR) dl.insert(i, degree[i]);

  std::vector<int64_t> w(n, 0);
  int64_t wflg = 1;
  std::vector<int> stampLp(n, -1), stampCmp(n, -1);
  int cmpTag = 0;
  std::vector<int> Lp, hashHead(n, -1), hashNext(n, -1), hashKeyOf(n, 0), touchedBuckets;

  int eliminated = 0;
  while (eliminated < nleft) {
    // --- pick the pivot of least approximate degree
    while (dl.mindeg < n && dl.head[dl.mindeg] < 0) ++dl.mindeg;
    const int p = dl.head[dl.mindeg];
    dl.remove(p, dl.mindeg);
    int nvpiv = nv[p];

    // --- form the new element L_p = (A_p U union of L_e, e in E_p) \ {p}
    Lp.clear();
    stampLp[p] = p;
    int degme = 0;
    for (int v : adjv[p]) {
      if (state[v] != ST_VAR || stampLp[v] == p) continue;
      stampLp[v] = p; Lp.push_back(v); degme += nv[v];
    }
    for (int e : adje[p]) {
      if (state[e] != ST_ELEM) continue;
      for (int v : elvars[e]) {
        if (state[v] != ST_VAR || stampLp[v] == p) continue;
        stampLp[v] = p; Lp.push_back(v); degme += nv[v];
      }
      state[e] = ST_DEAD_ELEM;  // absorbed into p
      std::vector<int>().swap(elvars[e]);
    }
    std::vector<int>().swap(adjv[p]);
    std::vector<int>().swap(adje[p]);
    state[p] = ST_ELEM;
    for (int v : Lp) dl.remove(v, degree[v]);

    // --- |L_e \ L_p| for every element adjacent to a member of L_p
    if (wflg > (int64_t)1 << 60) { std::fill(w.begin(), w.end(), 0); wflg = 1; }
    for (int i : Lp) {
      for (int e : adje[i]) {
        if (state[e] != ST_ELEM) continue;
        if (w[e] < wflg) w[e] = (int64_t)elemdeg[e] + wflg;
        w[e] -= nv[i];
      }
    }

    // --- degree update, list pruning, mass elimination, hashing
    touchedBuckets.clear();
    for (int i : Lp) {
      int64_t deg = 0;
      uint64_t hash = 0;
      auto& ei = adje[i];
      size_t k = 0;
      for (size_t q = 0; q < ei.size(); ++q) {
        int e = ei[q];
        if (state[e] != ST_ELEM) continue;
        int64_t dext = w[e] - wflg;
        if (dext > 0) { deg += dext; ei[k++] = e; hash += (uint64_t)e; }
        else { state[e] = ST_DEAD_ELEM; std::vector<int>().swap(elvars[e]); }  // aggressive absorption
      }
      ei.resize(k);
      auto& vi = adjv[i];
      k = 0;
      for (size_t q = 0; q < vi.size(); ++q) {
        int v = vi[q];
        if (state[v] != ST_VAR || stampLp[v] == p) continue;
        deg += nv[v]; vi[k++] = v; hash += (uint64_t)v;
      }
      vi.resize(k);
      if (deg == 0 && ei.empty()) {
        // mass elimination: i has become indistinguishable from the pivot
        state[i] = ST_ABSORBED;
        absorbed[p].push_back(i);
        nvpiv += nv[i];
        degme -= nv[i];
        continue;
      }
      ei.push_back(p);
      std::swap(ei.front(), ei.back());
      hash += (uint64_t)p;
      degree[i] = (int)std::min<int64_t>(degree[i], deg);
      int hk = (int)(hash % (uint64_t)n);
      hashKeyOf[i] = hk;
      if (hashHead[hk] < 0) touchedBuckets.push_back(hk);
      hashNext[i] = hashHead[hk];
      hashHead[hk] = i;
    }

    // --- supervariable detection among members of L_p with equal hash
    for (int hk : touchedBuckets) {
      for (int i = hashHead[hk]; i >= 0; i = hashNext[i]) {
        if (state[i] != ST_VAR) continue;
        bool marked = false;
        for (int j = hashNext[i]; j >= 0; j = hashNext[j]) {
          if (state[j] != ST_VAR) continue;
          if (adjv[i].size() != adjv[j].size() || adje[i].size() != adje[j].size()) continue;
          if (!marked) {
            ++cmpTag;
            for (int v : adjv[i]) stampCmp[v] = cmpTag;
            for (int e : adje[i]) stampCmp[e] = cmpTag;
            marked = true;
          }
          bool same = true;
          for (int v : adjv[j]) if (stampCmp[v] != cmpTag) { same = false; break; }
          if (same) for (int e : adje[j]) if (stampCmp[e] != cmpTag) { same = false; break; }
          if (!same) continue;
          // j joins supervariable i
          nv[i] += nv[j];
          nv[j] = 0;
          state[j] = ST_ABSORBED;
          absorbed[i].push_back(j);
          std::vector<int>().swap(adjv[j]);
          std::vector<int>().swap(adje[j]);
        }
      }
      hashHead[hk] = -1;
    }

    // --- finalise the element and re-insert the surviving members
    auto& lp = elvars[p];
    lp.clear();
    const int nleft_after = nleft - eliminated - nvpiv;
    for (int i : Lp) {
      if (state[i] != ST_VAR) continue;
      lp.push_back(i);
      int64_t d = (int64_t)degree[i] + degme - nv[i];
      d = std::min<int64_t>(d, (int64_t)nleft_after - nv[i]);
      if (d < 0) d = 0;
      degree[i] = (int)d;
      dl.insert(i, degree[i]);
    }
    elemdeg[p] = degme;
    if (lp.empty()) state[p] = ST_DEAD_ELEM;
    wflg += (int64_t)n + 1 + degme;  // keep stale w[] entries below the new flag
    eliminated += nvpiv;

    // --- emit p and everything ordered with it
    order.push_back(p);
    {
      // iterative traversal of the absorbed forest rooted at p
      size_t start = order.size() - 1;
      for (size_t q = start; q < order.size(); ++q) {
        int u = order[q];
        for (int c : absorbed[u]) order.push_back(c);
        std::vector<int>().swap(absorbed[u]);
      }
    }
  }
  for (int d : dense_nodes) order.push_back(d);
}

}  // namespace okkt
