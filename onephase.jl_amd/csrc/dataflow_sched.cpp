// Task queue of the dataflow factorisation of the big fronts of one level (host side, no HIP).
//
// A front of order f with k pivot columns is cut into blocks of 128 rows / columns: KB = ceil(k / 128) pivot blocks
// [128 b, min(128 (b + 1), k)) and then blocks of 128 from k on (the contribution block), TB blocks in all.  The blocked
// right-looking LDL^T of numeric.hip (k_big_diag2 -> k_big_trsm -> k_big_syrk per block column) becomes three kinds of tasks
// on the tiles (i, j), i >= j, of that grid:
//   D(q)             factor the diagonal tile (q, q), q < KB                         after every update of (q, q)
//   T(i, q)          W = A(i, q) L(q, q)^-T, L = W D^-1 for block row i > q          after D(q) and every update of (i, q)
//   U(i, j, q0, nq)  A(i, j) -= sum_{q0 <= q < q0 + nq} W(i, q) L(j, q)^T             after T(i, q), T(j, q) and the previous update of (i, j)
// A tile receives the panels in ascending order (the same sequence of operations per entry as the per-step kernels: bitwise the
// same factor); two panels per task (K = 256) wherever the column is not the very next one to be factored.
//
// The persistent kernel (dataflow.hip) pops tasks from ONE queue in order; a task spins until the tile states it depends on have
// been published.  Every dependency of a task lies earlier in the queue, so the launch cannot deadlock whatever the number of
// resident workgroups.  The ORDER decides the overlap: it is the start order of a list schedule simulated here with a crude time
// model (critical-path priority = leftmost target column first) -- the chain D -> T -> U -> D of the next block column is
// woven into the bulk updates of the previous ones at the positions where its inputs are expected to be ready.
#include <algorithm>
#include <cstdint>
#include <functional>
#include <queue>
#include <vector>

#include "numeric.h"

namespace okkt {

namespace {

struct FrontGrid {
  int f, k, KB, TB;
  int64_t offD, offT, offU;          // first task index of each kind
  std::vector<int> tq;               // [KB] prefix of the T tasks per panel
  std::vector<int64_t> uoff;         // [TB * TB] first U task of tile (i, j)
  int npanels(int j) const { return std::min(j, KB); }                      // panels tile column j receives
  int ngroups(int j, int G) const { return (npanels(j) + G - 1) / G; }
};

}  // namespace

void df_build_queue(const std::vector<DfFront>& fronts, int workers, int group, std::vector<DfTask>& out, double* model_us) {
  const int G = std::max(1, group);
  std::vector<FrontGrid> grids(fronts.size());
  struct Node { int front; int type; int i, j, q0, nq; int ndep; float dur; int64_t key; };
  std::vector<Node> nodes;
  for (size_t a = 0; a < fronts.size(); ++a) {
    FrontGrid& g = grids[a];
    g.f = fronts[a].f; g.k = fronts[a].k;
    g.KB = (g.k + 127) / 128;
    g.TB = g.KB + (g.f - g.k + 127) / 128;
    const int KB = g.KB, TB = g.TB;
    // priority: target column relative to the end of the pivot block (a front with a longer chain ahead of it goes first), then
    // the kind (D, T, U), then the row.  Smaller = more urgent.
    auto key = [&](int col, int kind, int row, int q0) { return ((int64_t)(col - KB + 4096) << 40) | ((int64_t)kind << 36) | ((int64_t)q0 << 20) | (int64_t)row; };
    g.offD = (int64_t)nodes.size();
    for (int q = 0; q < KB; ++q) {
      const int nb = std::min(128, g.k - 128 * q);
      nodes.push_back({(int)a, kDfD, q, q, q, 1, q > 0 ? 1 : 0, 5.0f + 2.1f * (float)((nb + 7) / 8), key(q, 0, q, q)});
    }
    g.offT = (int64_t)nodes.size();
    g.tq.assign(KB + 1, 0);
    for (int q = 0; q < KB; ++q) {
      g.tq[q + 1] = g.tq[q] + (TB - 1 - q);
      for (int i = q + 1; i < TB; ++i) nodes.push_back({(int)a, kDfT, i, q, q, 1, 1 + (q > 0 ? 1 : 0), 12.0f, key(q, 1, i, q)});
    }
    g.offU = (int64_t)nodes.size();
    g.uoff.assign((size_t)TB * TB, -1);
    for (int i = 0; i < TB; ++i)
      for (int j = 0; j <= i; ++j) {
        g.uoff[(size_t)i * TB + j] = (int64_t)nodes.size();
        const int np = g.npanels(j);
        for (int q0 = 0; q0 < np; q0 += G) {
          const int nq = std::min(G, np - q0);
          nodes.push_back({(int)a, kDfU, i, j, q0, nq, (i != j ? 2 : 1) + (q0 > 0 ? 1 : 0), 9.0f + 0.165f * 128.0f * (float)nq, key(j, 2, i, q0)});
        }
      }
  }
  // list schedule: `workers` identical workers, a ready task with the smallest key starts as soon as a worker is free
  typedef std::pair<int64_t, int64_t> KI;   // (key, node)
  std::priority_queue<KI, std::vector<KI>, std::greater<KI>> ready;
  typedef std::pair<double, int64_t> TI;    // (finish time, node)
  std::priority_queue<TI, std::vector<TI>, std::greater<TI>> running;
  for (int64_t x = 0; x < (int64_t)nodes.size(); ++x) if (nodes[x].ndep == 0) ready.push({nodes[x].key, x});
  out.clear();
  out.reserve(nodes.size());
  double now = 0;
  int idle = std::max(1, workers);
  auto release = [&](int64_t x) { if (--nodes[x].ndep == 0) ready.push({nodes[x].key, x}); };
  while (!ready.empty() || !running.empty()) {
    while (idle > 0 && !ready.empty()) {
      const int64_t x = ready.top().second;
      ready.pop();
      const Node& nd = nodes[x];
      out.push_back({fronts[nd.front].s, nd.type | (nd.nq << 8), nd.i | (nd.j << 16), nd.q0});
      running.push({now + nd.dur, x});
      --idle;
    }
    if (running.empty()) break;
    const int64_t x = running.top().second;
    now = running.top().first;
    running.pop();
    ++idle;
    const Node nd = nodes[x];
    const FrontGrid& g = grids[nd.front];
    const int KB = g.KB, TB = g.TB;
    if (nd.type == kDfD) {
      const int q = nd.i;
      for (int i = q + 1; i < TB; ++i) release(g.offT + g.tq[q] + (i - q - 1));
    } else if (nd.type == kDfT) {
      // T(i, q): the update groups whose LAST panel is q and that read block row i as the row operand (tiles (i, j), q < j <= i)
      // or as the column operand (tiles (i2, i), i2 >= i)
      const int i = nd.i, q = nd.j;
      auto group_of = [&](int j) -> int64_t {   // index (within tile column j) of the group that ends with panel q, or -1
        const int np = g.npanels(j);
        if (q >= np) return -1;
        const int gq = q / G;
        const int last = std::min(gq * G + G, np) - 1;
        return last == q ? gq : -1;
      };
      for (int j = q + 1; j <= i; ++j) { const int64_t gq = group_of(j); if (gq >= 0) release(g.uoff[(size_t)i * TB + j] + gq); }
      { const int64_t gq = group_of(i); if (gq >= 0) for (int i2 = i + 1; i2 < TB; ++i2) release(g.uoff[(size_t)i2 * TB + i] + gq); }
    } else {
      const int i = nd.i, j = nd.j;
      const int np = g.npanels(j);
      if (nd.q0 + nd.nq < np) release(g.uoff[(size_t)i * TB + j] + nd.q0 / G + 1);
      else if (j < KB) release(i == j ? g.offD + j : g.offT + g.tq[j] + (i - j - 1));
    }
  }
  if (model_us) *model_us = now;
}

}  // namespace okkt
