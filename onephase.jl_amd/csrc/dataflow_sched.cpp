// Task queue of the dataflow factorisation of the big fronts of one level (host side, no HIP).
//
// A front of order f with k pivot columns is cut into blocks of 128 rows / columns: KB = ceil(k / 128) pivot blocks
// [128 b, min(128 (b + 1), k)) and then blocks of 128 from k on (the contribution block), TB blocks in all.  The blocked
// right-looking LDL^T of numeric.hip (k_big_diag2 -> k_big_trsm -> k_big_syrk per block column) becomes tasks on the tiles
// (i, j), i >= j, of that grid:
//   D(q)             factor the diagonal tile (q, q), q < KB                         after every update of (q, q)
//   T(i, q)          W = A(i, q) L(q, q)^-T, L = W D^-1 for block row i > q          after D(q) and every update of (i, q)
//   U(i, j, q0, nq)  A(i, j) -= sum_{q0 <= q < q0 + nq} W(i, q) L(j, q)^T             after T(i, q), T(j, q) and the previous update of (i, j)
//   TU(q)            T(q + 1, q) and U(q + 1, q + 1, q, 1) in one task (q + 1 < KB): the two steps between the diagonal blocks of
//                    consecutive block columns -- the critical path -- without a hand-off and without a trip through memory
//   TL(i, q)         (fuse_tl) T(i, q) with the LAST update of its tile, U(i, q, q - 1, 1), inside the task, for every block row below TU's and
//                    q >= 1: the row's per-column chain T -> single-panel update -> T (two tasks and two hand-offs per block column: as
//                    long as the 63 us between two diagonal blocks, so every row held the chain back) becomes one task whose update runs
//                    BEFORE D(q) has arrived (released when D(q) starts, like TU)
//   TA(q)            the upper 64 rows of TU(q)'s block row, when it has more than 64 (split_tu): both steps run at the FP64 matrix rate
//                    of one CU, two workers halve them.  Emitted right before TU(q), which waits for TA's two states inside the task
// A tile receives the panels in ascending order (the same sequence of operations per entry as the per-step kernels: bitwise the
// same factor).  Panels are applied `group` at a time (K = 128 * group) except that the LAST panel of a pivot column comes alone:
// it is the one the next diagonal block (or panel tile) waits for.
//
// The persistent kernel (dataflow.hip) pops tasks from ONE queue in order; a task spins until the tile states it depends on have
// been published.  Every dependency of a task lies earlier in the queue, so the launch cannot deadlock whatever the number of
// resident workgroups.  The ORDER decides the overlap: it is the start order of a list schedule simulated here with a crude time
// model (critical-path priority = leftmost target column first) -- the chain D -> TU -> D of the next block column is woven
// into the bulk updates of the previous ones at the positions where its inputs are expected to be ready.  TU(q) is started
// together with D(q): its worker has the rows of its tiles in flight while the diagonal block is being factored.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <queue>
#include <unordered_map>
#include <vector>

#include "numeric.h"

namespace okkt {

namespace {

constexpr int kLonePanels = 2;

struct FrontGrid {
  int f, k, KB, TB;
  int64_t offD, offT, offTU, offU;   // first task index of each kind
  std::vector<int> tq;               // [KB] prefix of the T tasks per panel
  std::vector<std::vector<std::vector<int64_t>>> unode;   // [column j][group c][row i]: the U task that holds tile (i, j) for that group
  std::vector<std::vector<int>> gstart;   // [TB] first panel of every update group of tile column j, plus the end
  int npanels(int j) const { return std::min(j, KB); }                      // panels tile column j receives
  // index of the group of column j that ENDS with panel q, or -1
  int group_ending(int j, int q) const {
    const std::vector<int>& g = gstart[j];
    for (size_t a = 0; a + 1 < g.size(); ++a) if (g[a + 1] - 1 == q) return (int)a;
    return -1;
  }
};

}  // namespace

void df_build_queue(const std::vector<DfFront>& fronts, int workers, int group, int rows_per_task, bool fuse_d, bool split_tu, std::vector<DfTask>& out, double* model_us,
                    bool fuse_tl, bool lockstep, bool multi_rows) {
  if (lockstep) split_tu = false;
  if (!multi_rows) rows_per_task = 1;      // (a library whose update role carries one tile per task: dataflow.hip, kDfMulti)      // one worker follows D(q) block by block (dataflow.hip, df_tu_lock): nothing left to split
  const int G = std::max(1, group);
  const int RT = std::max(1, std::min(rows_per_task, 8));
  // Time model (us, measured on MI355X with one workgroup per CU: D 39-45, T 25-32, U 25-30 at K = 128 and 45-52 at K = 256, TU 40
  // behind its D).  The chain tasks D and TU enter the model SHORTER than they are (chain_scale): a chain task that sits in the
  // queue behind bulk tasks of the same readiness is popped late and the whole chain slips (priority inversion, seen as 75-105 us
  // between diagonal blocks instead of 45 in the bulk-bound phase of the S-metric root); one that is popped early only costs a
  // waiting workgroup.
  static const float chain_scale = getenv("OKKT_DF_MODEL_CHAIN") ? (float)atof(getenv("OKKT_DF_MODEL_CHAIN")) : 0.6f;
  static const float bulk_scale = getenv("OKKT_DF_MODEL_BULK") ? (float)atof(getenv("OKKT_DF_MODEL_BULK")) : 1.0f;
  std::vector<FrontGrid> grids(fronts.size());
  struct Node { int front; int type; int i, j, q0, nq; int ndep; float dur; int64_t key; int rows; int64_t fwd; };      // type -1: a placeholder (an update that runs inside the task `fwd`)
  std::vector<Node> nodes;
  for (size_t a = 0; a < fronts.size(); ++a) {
    FrontGrid& g = grids[a];
    g.f = fronts[a].f; g.k = fronts[a].k;
    g.KB = (g.k + 127) / 128;
    g.TB = g.KB + (g.f - g.k + 127) / 128;
    const int KB = g.KB, TB = g.TB;
    // update groups of every tile column
    g.gstart.assign(TB, std::vector<int>());
    for (int j = 0; j < TB; ++j) {
      const int np = g.npanels(j);
      // a pivot column takes its last panels alone (K = 128): the last one is what the next diagonal block / panel tile waits
      // for, the one before it becomes available one step earlier and must be done by then; everything older comes `group` panels
      // at a time (K = 512 at the default: 64 us of MFMA work for the 20 us a task pays around it)
      const int lone = j < KB ? std::min(np, kLonePanels) : 0;
      const int grouped = np - lone;
      std::vector<int>& gs = g.gstart[j];
      for (int q0 = 0; q0 < grouped; q0 += G) gs.push_back(q0);
      for (int q0 = grouped; q0 < np; ++q0) gs.push_back(q0);
      gs.push_back(np);
    }
    // priority: target column relative to the end of the pivot block (a front with a longer chain ahead of it goes first), then
    // the kind (D, TU, T, U), then the row.  Smaller = more urgent.
    auto key = [&](int col, int kind, int row, int q0) { return ((int64_t)(col - KB + 4096) << 40) | ((int64_t)kind << 36) | ((int64_t)q0 << 20) | (int64_t)row; };
    g.offD = (int64_t)nodes.size();
    for (int q = 0; q < KB; ++q) {
      const int nb = std::min(128, g.k - 128 * q);
      nodes.push_back({(int)a, kDfD, q, q, q, 1, q > 0 ? 1 : 0, chain_scale * (5.0f + 2.3f * (float)((nb + 7) / 8)), key(q, 0, q, q), 1, -1});
    }
    g.offTU = (int64_t)nodes.size();
    for (int q = 0; q + 1 < KB; ++q)     // needs D(q) [released when D(q) STARTS], the updates of (q + 1, q) and the earlier updates of (q + 1, q + 1)
      nodes.push_back({(int)a, kDfTU, q + 1, q, q, 1, 1 + (q > 0 ? 2 : 0), 0.0f, key(q, 1, q + 1, q), 1, -1});
    g.offT = (int64_t)nodes.size();
    g.tq.assign(KB + 1, 0);
    for (int q = 0; q < KB; ++q) {
      const int first = q + 1 < KB ? q + 2 : q + 1;             // block row q + 1 belongs to TU(q)
      g.tq[q + 1] = g.tq[q] + std::max(TB - first, 0);
      // fuse_tl: from panel 1 on the panel tile carries the last update of its tile (TL): it waits for that update's operands (the
      // placeholder below forwards them) and for the START of D(q)
      for (int i = first; i < TB; ++i)
        nodes.push_back({(int)a, (fuse_tl && q > 0) ? kDfTL : kDfT, i, q, q, 1, 1 + (q > 0 ? 1 : 0), bulk_scale * 28.0f, key(q, 2, i, q), 1, -1});
    }
    // Update tasks.  The groups of a column that are not the lone last panel of a pivot column are bulk work: the tiles below the
    // diagonal tile are taken `rows_per_task` at a time (one pop, one wait, one acquire and one drain per task, and the C tile of
    // the next row block is in flight while the current one is computed: dataflow.hip, df_syrk_tiles).  The last panel of a pivot
    // column stays one tile per task: the panel tile below it (or the next diagonal block) waits for exactly that tile.
    g.unode.assign(TB, std::vector<std::vector<int64_t>>());
    for (int j = 0; j < TB; ++j) {
      const std::vector<int>& gs = g.gstart[j];
      const int ng = (int)gs.size() - 1;
      g.unode[j].assign(ng, std::vector<int64_t>(TB, -1));
      for (int c = 0; c < ng; ++c) {
        const int q0 = gs[c], nq = gs[c + 1] - gs[c];
        const bool lone_last = j < KB && c + 1 == ng;
        // Row tiles per bulk task.  Two row tiles per task cost 17 % less per tile (one pop, one wait, one drain; the next C tile in
        // flight) and lost it all in the schedule when every bulk task carried them (round 4: 19.5 ms against 18.85) or the tasks far
        // from the diagonal did (round 5: 17.0 - 17.8 against 16.8): coarser tasks starve the tails.  Round 6 gates it in TIME, not in
        // space: the tasks of a panel group carry `rows_big` tiles only while the front still has `rows_ahead` block columns to go
        // behind the group and is long enough to have an update-bound phase at all (`rows_minkb` pivot blocks) -- the 96 %-busy phase of
        // the metric workload's root and of the front below it; every tail keeps single tiles.
        static const int rows_big = getenv("OKKT_DF_ROWS_BIG") ? std::max(1, std::min(atoi(getenv("OKKT_DF_ROWS_BIG")), 8)) : 1;
        static const int rows_minkb = getenv("OKKT_DF_ROWS_MINKB") ? atoi(getenv("OKKT_DF_ROWS_MINKB")) : 32;
        static const int rows_ahead = getenv("OKKT_DF_ROWS_AHEAD") ? atoi(getenv("OKKT_DF_ROWS_AHEAD")) : 24;
        // ... and in SPACE, by the distance of the tile column from the group's last panel: the chain reaches column j that many
        // steps later, and a task that feeds it must be done by then -- a pair of tiles takes 157 us where one takes 88, and the pairs
        // next to the panel made the chain wait (103 us per block column in a phase where half the workers were idle; task log of the
        // metric workload's root).  The contribution block's columns (j >= KB) are nobody's input inside this launch.
        static const int rows_coldist = getenv("OKKT_DF_ROWS_COLDIST") ? atoi(getenv("OKKT_DF_ROWS_COLDIST")) : 4;
        const int q_last = gs[c + 1] - 1;                                   // last panel of the group
        const bool far = j >= KB || j - q_last >= rows_coldist;
        const int RG = (multi_rows && rows_big > 1 && far && KB >= rows_minkb && KB - 1 - q_last >= rows_ahead) ? std::max(RT, rows_big) : RT;
        const int R = (j < KB && c + kLonePanels >= ng) ? 1 : RG;
        for (int i = j; i < TB;) {
          const int rows = i == j ? 1 : std::min(R, TB - i);
          const bool in_tu = i == j && lone_last;                  // the last panel of a diagonal pivot tile: part of TU(j - 1)
          const bool in_tl = fuse_tl && lone_last && i > j && !(i == j + 1 && j + 1 < KB);      // ... of a tile below the TU row: part of TL(i, j)
          const int ndep = (i != j ? rows + 1 : 1) + (c > 0 ? 1 : 0);
          for (int r = 0; r < rows; ++r) g.unode[j][c][i + r] = (int64_t)nodes.size();
          nodes.push_back({(int)a, in_tu ? -1 : (in_tl ? -2 : kDfU), i, j, q0, nq, ndep,
                           bulk_scale * (6.0f + 0.16f * 128.0f * (float)nq * (float)rows + 4.0f * (float)(rows - 1)), key(j, 3, i, q0), rows,
                           in_tl ? g.offT + g.tq[j] + (i - (j + 1 < KB ? j + 2 : j + 1)) : -1});      // in_tl (type -2): once its operands are there it releases TL(i, j)
          i += rows;
        }
      }
    }
  }
  auto t_index = [&](const FrontGrid& g, int i, int q) -> int64_t {       // T(i, q) or, for i == q + 1 < KB, TU(q)
    if (q + 1 < g.KB) return i == q + 1 ? g.offTU + q : g.offT + g.tq[q] + (i - q - 2);
    return g.offT + g.tq[q] + (i - q - 1);
  };
  // list schedule: `workers` identical workers, a ready task with the smallest key starts as soon as a worker is free
  typedef std::pair<int64_t, int64_t> KI;   // (key, node)
  std::priority_queue<KI, std::vector<KI>, std::greater<KI>> ready;
  typedef std::pair<double, int64_t> TI;    // (finish time, node)
  std::priority_queue<TI, std::vector<TI>, std::greater<TI>> running;
  for (int64_t x = 0; x < (int64_t)nodes.size(); ++x) if (nodes[x].type >= 0 && nodes[x].ndep == 0) ready.push({nodes[x].key, x});
  out.clear();
  out.reserve(nodes.size());
  double now = 0;
  int idle = std::max(1, workers);
  std::function<void(int64_t)> release = [&](int64_t x) {
    if (--nodes[x].ndep != 0) return;
    if (nodes[x].type == -2) release(nodes[x].fwd);      // an update inside TL: its operands are what the task waits for
    else ready.push({nodes[x].key, x});
  };
  std::vector<std::vector<double>> dend(fronts.size());      // [front][q]: end of D(q) in the model (TL tasks start before it)
  for (size_t a = 0; a < fronts.size(); ++a) dend[a].assign(grids[a].KB, 0.0);
  while (!ready.empty() || !running.empty()) {
    while (idle > 0 && !ready.empty()) {
      const int64_t x = ready.top().second;
      ready.pop();
      Node& nd = nodes[x];
      // fuse_d: D(q), q >= 1, is carried out by the worker of TU(q - 1) right behind the update of its tile (no task of its own; it stays
      // in this simulation, where it starts on some worker the moment TU(q - 1) ends -- the same thing for the model)
      if (nd.type == kDfTU) {
        // a block row of more than 64 rows is split between two workers: TA (the upper 64 rows) goes into the queue right before
        // TU (the rest, and everything behind the tile update) -- TU waits for TA's two states, so TA has to be the earlier pop
        const FrontGrid& gg = grids[nd.front];
        const int lo = std::min(nd.i * 128, gg.k), hi = std::min(lo + 128, gg.k);      // nd.i = q + 1 < KB: a pivot block
        const bool split = split_tu && hi - lo > 64;
        if (split) out.push_back({fronts[nd.front].s, kDfTA | (1 << 8) | (1 << 16), nd.i | (nd.j << 16), nd.q0});
        out.push_back({fronts[nd.front].s, kDfTU | (((fuse_d ? 2 : 1) | (split ? 4 : 0) | (lockstep ? 8 : 0)) << 8) | (nd.rows << 16), nd.i | (nd.j << 16), nd.q0});
      } else if (!(fuse_d && nd.type == kDfD && nd.i > 0))
        out.push_back({fronts[nd.front].s, nd.type | ((nd.nq | ((lockstep && nd.type == kDfD) ? 8 : 0)) << 8) | (nd.rows << 16), nd.i | (nd.j << 16), nd.q0});      // D with bit 8: report the finished 32-column blocks
      if (nd.type == kDfTL)      // the update (14 us) runs ahead of D(q); the solve, the stores and the hand-off behind it
        nd.dur = (float)(std::max(now + 14.0 * bulk_scale, dend[nd.front][nd.j]) + 22.0 * bulk_scale - now);
      running.push({now + nd.dur, x});
      --idle;
      if (nd.type == kDfD) {
        dend[nd.front][nd.i] = now + nd.dur;
        if (nd.i + 1 < grids[nd.front].KB) {      // TU(q) starts beside D(q) and ends 24 us behind it
          const int64_t tu = grids[nd.front].offTU + nd.i;
          nodes[tu].dur = nd.dur + chain_scale * (lockstep ? 14.0f : 40.0f);      // lockstep: only the last 32-column block is left behind D(q)
          release(tu);
        }
        if (fuse_tl && nd.i > 0) {                // ... and so do the TL tasks of the block rows below
          const FrontGrid& gg = grids[nd.front];
          const int q = nd.i;
          for (int i = (q + 1 < gg.KB ? q + 2 : q + 1); i < gg.TB; ++i) {
            const int64_t x2 = gg.offT + gg.tq[q] + (i - (q + 1 < gg.KB ? q + 2 : q + 1));
            if (nodes[x2].type == kDfTL) release(x2);
          }
        }
      }
    }
    if (running.empty()) break;
    const int64_t x = running.top().second;
    now = running.top().first;
    running.pop();
    const Node nd = nodes[x];
    ++idle;
    const FrontGrid& g = grids[nd.front];
    const int KB = g.KB, TB = g.TB;
    if (nd.type == kDfD) {
      const int q = nd.i;
      for (int i = (q + 1 < KB ? q + 2 : q + 1); i < TB; ++i)
        if (nodes[t_index(g, i, q)].type != kDfTL) release(t_index(g, i, q));      // (the TL tasks were released when D(q) started)
    } else if (nd.type == kDfT || nd.type == kDfTU || nd.type == kDfTL) {
      // block row i of panel q is done: the update groups whose LAST panel is q and that read block row i as the row operand
      // (tiles (i, j), q < j <= i) or as the column operand (tiles (i2, i), i2 > i: once per task)
      const int i = nd.i, q = nd.j;
      for (int j = q + 1; j <= i; ++j) {
        if (nd.type == kDfTU && j == i) continue;                 // its own diagonal tile: updated inside the task
        const int gq = g.group_ending(j, q);
        if (gq >= 0) release(g.unode[j][gq][i]);
      }
      {
        const int gq = g.group_ending(i, q);
        if (gq >= 0) {
          int64_t last = -1;
          for (int i2 = i + 1; i2 < TB; ++i2) { const int64_t x2 = g.unode[i][gq][i2]; if (x2 != last) { release(x2); last = x2; } }
        }
      }
      if (nd.type == kDfTU) release(g.offD + i);                  // tile (q + 1, q + 1) has received panel q
    } else {
      // an update task: the next group of its tiles (one task with the same rows, or one task per tile when the next group is the
      // lone last panel of a pivot column), or what waits for the finished tiles
      const int j = nd.j;
      const int ng = (int)g.gstart[j].size() - 1;
      const int c = g.group_ending(j, nd.q0 + nd.nq - 1);
      if (c + 1 < ng) {
        int64_t last = -1;
        for (int r = 0; r < nd.rows; ++r) {
          const int64_t x2 = g.unode[j][c + 1][nd.i + r];
          if (x2 == last) continue;
          last = x2;
          if (nodes[x2].type != -1) release(x2); else release(g.offTU + (j - 1));      // ... the one inside TU(j - 1)
        }
      } else if (j < KB) {
        for (int r = 0; r < nd.rows; ++r) release(nd.i + r == j ? g.offD + j : t_index(g, nd.i + r, j));
      }
    }
  }
  if (model_us) *model_us = now;
}

}  // namespace okkt
