// Host-side sanitizer run (make asan): the orderings, the symbolic analysis and the dataflow queue builder under AddressSanitizer and
// UndefinedBehaviorSanitizer on small synthetic patterns -- a banded KKT (level-structure dissection), a 3-D grid, a small-world graph
// (multilevel dissection with its helper threads) and an arrow matrix.  No device code, no HIP call.  SURVEY.md section 5 (host-side
// sanitizer build); GPU sanitizers are not available on this pool.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "numeric.h"
#include "symbolic.h"

using namespace okkt;

static void csc_lower(int n, const std::set<std::pair<int, int>>& e, std::vector<int64_t>& cp, std::vector<int64_t>& ri) {
  std::vector<std::vector<int>> cols(n);
  for (auto& p : e) cols[std::min(p.first, p.second)].push_back(std::max(p.first, p.second));
  cp.assign(n + 1, 0); ri.clear();
  for (int j = 0; j < n; ++j) { ri.push_back(j); for (int r : cols[j]) if (r != j) ri.push_back(r); cp[j + 1] = (int64_t)ri.size(); }
}

static int run(const char* name, int n, const std::set<std::pair<int, int>>& e, int ordering) {
  std::vector<int64_t> cp, ri;
  csc_lower(n, e, cp, ri);
  SymbolicOptions o;
  o.ordering = ordering;
  Symbolic S;
  const std::string err = analyze_pattern(n, cp.data(), ri.data(), 0, o, nullptr, S);
  if (!err.empty()) { fprintf(stderr, "%s (ordering %d): %s\n", name, ordering, err.c_str()); return 1; }
  // the dataflow queues of the three largest fronts as one level, both forms
  std::vector<DfFront> fronts;
  for (int s = 0; s < (int)S.sn_col0.size() - 1 && fronts.size() < 3; ++s) {
    const int k = S.sn_col0[s + 1] - S.sn_col0[s], f = (int)(S.row_ptr[s + 1] - S.row_ptr[s]);
    if (f > 128) fronts.push_back({s, f, k});
  }
  size_t ntask = 0;
  if (!fronts.empty()) {
    std::vector<DfTask> q;
    double model = 0;
    df_build_queue(fronts, 64, 4, 1, true, true, q, &model, true);
    ntask = q.size();
    df_build_queue(fronts, 96, 2, 2, false, false, q, &model, false);
  }
  printf("%-12s ordering %d: n %d nnz(L) %lld supernodes %zu, %zu dataflow tasks\n", name, ordering, n, (long long)S.nnzL, S.sn_col0.size() - 1, ntask);
  return 0;
}

int main() {
  int bad = 0;
  std::mt19937 rng(5);
  {   // banded KKT with one dense row
    const int n = 3000;
    std::set<std::pair<int, int>> e;
    for (int i = 0; i < n; ++i) for (int d = 1; d <= 3; ++d) if (i + d < n) e.insert({i, i + d});
    for (int i = 0; i < n; i += 2) e.insert({i, n - 1});
    for (int o : {0, 3, 4}) bad += run("banded", n, e, o);
  }
  {   // 3-D grid 14^3
    const int g = 14, n = g * g * g;
    std::set<std::pair<int, int>> e;
    auto id = [&](int x, int y, int z) { return (x * g + y) * g + z; };
    for (int x = 0; x < g; ++x) for (int y = 0; y < g; ++y) for (int z = 0; z < g; ++z) {
      if (x + 1 < g) e.insert({id(x, y, z), id(x + 1, y, z)});
      if (y + 1 < g) e.insert({id(x, y, z), id(x, y + 1, z)});
      if (z + 1 < g) e.insert({id(x, y, z), id(x, y, z + 1)});
    }
    for (int o : {0, 3, 4, 5}) bad += run("grid", n, e, o);
  }
  {   // small world: a band plus random long edges (the shape of the metric workload), large enough for the threaded dissection
    const int n = 12000;
    std::set<std::pair<int, int>> e;
    for (int i = 0; i < n; ++i) for (int d = 1; d <= 12; d += 3) if (i + d < n) e.insert({i, i + d});
    for (int t = 0; t < n / 4; ++t) { const int a = (int)(rng() % n), b = (int)(rng() % n); if (a != b) e.insert({a, b}); }
    for (int o : {0, 5}) bad += run("small-world", n, e, o);
  }
  {   // the same shape at 24 000: the sizes from which the cut statistics and the induced subgraphs go to helper threads as well
    const int n = 24000;
    std::set<std::pair<int, int>> e;
    for (int i = 0; i < n; ++i) for (int d = 1; d <= 12; d += 3) if (i + d < n) e.insert({i, i + d});
    for (int t = 0; t < n / 4; ++t) { const int a = (int)(rng() % n), b = (int)(rng() % n); if (a != b) e.insert({a, b}); }
    bad += run("small-world-24k", n, e, 5);
  }
  {   // arrow
    const int n = 600;
    std::set<std::pair<int, int>> e;
    for (int i = 0; i + 1 < n; ++i) { e.insert({i, n - 1}); e.insert({i, i + 1}); }
    for (int o : {0, 1, 3}) bad += run("arrow", n, e, o);
  }
  if (bad) { fprintf(stderr, "%d case(s) failed\n", bad); return 1; }
  printf("asan driver: all cases clean\n");
  return 0;
}
