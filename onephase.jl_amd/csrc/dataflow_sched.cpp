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

void df_build_queue(const std::vector<DfFront>& fronts, int workers, int group, int rows_per_task, bool fuse_d, bool split_tu, std::vector<DfTask>& out, double* model_us) {
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
  struct Node { int front; int type; int i, j, q0, nq; int ndep; float dur; int64_t key; int rows; };
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
      // OKKT_DF_TAPER = t > 0: the group right in front of the lone panels holds at most t panels and the full groups are aligned to
      // IT (the short group comes first): behind D(q - 3) the tile (q + 1, q) then needs T, a K = 128 t update and the two lone ones
      // (28 + 47 + 28 + 28 us at t = 2) within three chain steps (189 us) instead of T + a K = 512 update + the lone ones (181 us)
      static const int taper_env = getenv("OKKT_DF_TAPER") ? atoi(getenv("OKKT_DF_TAPER")) : 0;
      const int taper = std::min(taper_env, G - 1);      // (a group of G panels is not a taper)
      if (taper > 0 && j < KB && grouped > taper) {
        const int head_part = (grouped - taper) % G;
        if (head_part > 0) gs.push_back(0);
        for (int q0 = head_part; q0 < grouped - taper; q0 += G) gs.push_back(q0);
        gs.push_back(grouped - taper);
      } else {
        for (int q0 = 0; q0 < grouped; q0 += G) gs.push_back(q0);
      }
      for (int q0 = grouped; q0 < np; ++q0) gs.push_back(q0);
      gs.push_back(np);
    }
    // priority: target column relative to the end of the pivot block (a front with a longer chain ahead of it goes first), then
    // the kind (D, TU, T, U), then the row.  Smaller = more urgent.
    auto key = [&](int col, int kind, int row, int q0) { return ((int64_t)(col - KB + 4096) << 40) | ((int64_t)kind << 36) | ((int64_t)q0 << 20) | (int64_t)row; };
    g.offD = (int64_t)nodes.size();
    for (int q = 0; q < KB; ++q) {
      const int nb = std::min(128, g.k - 128 * q);
      nodes.push_back({(int)a, kDfD, q, q, q, 1, q > 0 ? 1 : 0, chain_scale * (5.0f + 2.3f * (float)((nb + 7) / 8)), key(q, 0, q, q), 1});
    }
    g.offTU = (int64_t)nodes.size();
    for (int q = 0; q + 1 < KB; ++q)     // needs D(q) [released when D(q) STARTS], the updates of (q + 1, q) and the earlier updates of (q + 1, q + 1)
      nodes.push_back({(int)a, kDfTU, q + 1, q, q, 1, 1 + (q > 0 ? 2 : 0), 0.0f, key(q, 1, q + 1, q), 1});
    g.offT = (int64_t)nodes.size();
    g.tq.assign(KB + 1, 0);
    for (int q = 0; q < KB; ++q) {
      const int first = q + 1 < KB ? q + 2 : q + 1;             // block row q + 1 belongs to TU(q)
      g.tq[q + 1] = g.tq[q] + std::max(TB - first, 0);
      for (int i = first; i < TB; ++i) nodes.push_back({(int)a, kDfT, i, q, q, 1, 1 + (q > 0 ? 1 : 0), bulk_scale * 28.0f, key(q, 2, i, q), 1});
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
        const int R = (j < KB && c + kLonePanels >= ng) ? 1 : RT;
        for (int i = j; i < TB;) {
          const int rows = i == j ? 1 : std::min(R, TB - i);
          const bool in_tu = i == j && lone_last;                  // the last panel of a diagonal pivot tile: part of TU(j - 1)
          const int ndep = (i != j ? rows + 1 : 1) + (c > 0 ? 1 : 0);
          for (int r = 0; r < rows; ++r) g.unode[j][c][i + r] = (int64_t)nodes.size();
          nodes.push_back({(int)a, in_tu ? -1 : kDfU, i, j, q0, nq, ndep, bulk_scale * (6.0f + 0.16f * 128.0f * (float)nq * (float)rows + 4.0f * (float)(rows - 1)), key(j, 3, i, q0), rows});
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
  auto release = [&](int64_t x) { if (--nodes[x].ndep == 0) ready.push({nodes[x].key, x}); };
  // only with D(q + 1) inside TU(q): as a task of its own it would come BEHIND the panel tile that waits for it
  const bool early_feeder = fuse_d && rows_per_task == 1 && getenv("OKKT_DF_HOIST") && atoi(getenv("OKKT_DF_HOIST")) != 0;
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
        out.push_back({fronts[nd.front].s, kDfTU | (((fuse_d ? 2 : 1) | (split ? 4 : 0)) << 8) | (nd.rows << 16), nd.i | (nd.j << 16), nd.q0});
      } else if (!(fuse_d && nd.type == kDfD && nd.i > 0))
        out.push_back({fronts[nd.front].s, nd.type | (nd.nq << 8) | (nd.rows << 16), nd.i | (nd.j << 16), nd.q0});
      running.push({now + nd.dur, x});
      --idle;
      if (nd.type == kDfD && nd.i + 1 < grids[nd.front].KB) {      // TU(q) starts beside D(q) and ends 24 us behind it
        const int64_t tu = grids[nd.front].offTU + nd.i;
        nodes[tu].dur = nd.dur + chain_scale * 40.0f;
        release(tu);
        // ... and the panel tile that feeds the NEXT chain step, T(q + 3, q + 1), counts D(q + 1) -- which TU(q) carries -- as done
        // already: it goes into the queue a chain step ahead of the model's time and waits there (the model's clock runs behind the
        // real chain while updates are plentiful: the feeder was popped 10 - 40 us after it could have started).  T(2, 0) beside D(0).
        if (early_feeder) {
          const FrontGrid& gg = grids[nd.front];
          if (nd.i == 0 && 2 < gg.TB) release(t_index(gg, 2, 0));
          if (nd.i + 2 < gg.KB && nd.i + 3 < gg.TB) release(t_index(gg, nd.i + 3, nd.i + 1));
        }
      }
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
      for (int i = (q + 1 < KB ? q + 2 : q + 1); i < TB; ++i) {
        if (early_feeder && i == q + 2 && q + 1 < KB && (q == 0 || q - 1 + 2 < KB)) continue;      // released when D(q - 1) (D(0)) was dispatched
        release(t_index(g, i, q));
      }
    } else if (nd.type == kDfT || nd.type == kDfTU) {
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
          if (nodes[x2].type >= 0) release(x2); else release(g.offTU + (j - 1));      // ... the one inside TU(j - 1)
        }
      } else if (j < KB) {
        for (int r = 0; r < nd.rows; ++r) release(nd.i + r == j ? g.offD + j : t_index(g, nd.i + r, j));
      }
    }
  }
  if (model_us) *model_us = now;

  // ---- the feeders of the chain, hoisted ---------------------------------------------------------------------------------------
  // TU(q) needs the tiles (q + 1, q) and (q + 1, q + 1) with panel q - 1 applied: two single-panel update tasks that wait for
  // T(q + 1, q - 1).  The simulation dispatches them when T(q + 1, q - 1) ends, i.e. BEHIND everything else that became ready in
  // those 28 us (the other 50 panel tiles of the step and the updates they release: 280 tasks at the root of the metric workload).
  // With every worker busy the queue advances 2.8 tasks per us, so the two tasks -- and TU(q) behind them -- were popped 30 - 100 us
  // after they could have run: the root's diagonal blocks came every 87 us instead of every 63 while updates were plentiful, and
  // its last 17 block columns were left over as a chain-bound tail.  With OKKT_DF_HOIST=1 they are moved up to right behind
  // T(q + 1, q - 1), TA(q) and TU(q) right behind them (their workers wait).  A task is never moved in front of one it depends on (the
  // dependencies are looked up; tests/test_dataflow_queue.py replays the result).  MEASURED: the two tasks then start the moment their
  // panel tile is done -- and the panel tile T(q + 1, q - 1) is what comes late (10 - 45 us behind D(q - 1)); moved up as well
  // (OKKT_DF_HOIST_LEAD), it waits for ITS last update, U(q + 1, q - 1; q - 2): the chain is gated by a widening cone of single-panel
  // tasks next to the diagonal, each of which queues behind bulk work, and the root takes 6.4 ms either way.  Off.  What the cone
  // needs is a priority the in-order queue cannot give it: a second queue for the tasks within a few block rows of the chain.
  static const bool hoist = getenv("OKKT_DF_HOIST") && atoi(getenv("OKKT_DF_HOIST")) != 0;      // experiment, off: see below
  if (hoist && rows_per_task == 1) {
    auto key_of = [](const DfTask& t) {
      const int type = t.type_nq & 255, nq = (t.type_nq >> 8) & 255, i = t.ij & 0xffff, j = t.ij >> 16;
      const int endp = type == kDfU ? t.q0 + nq : j + 1;          // panels applied to tile (i, j) once the task is done
      return ((int64_t)type << 48) | ((int64_t)i << 32) | ((int64_t)j << 16) | (int64_t)endp;
    };
    for (size_t a = 0; a < fronts.size(); ++a) {
      const FrontGrid& g = grids[a];
      if (g.KB < 3) continue;
      const int sfront = fronts[a].s;
      // positions by key for this front, kept up to date across the moves (a move shifts a short range of the queue)
      std::unordered_map<int64_t, int64_t> where;
      where.reserve(out.size() / fronts.size() * 2 + 16);
      for (int64_t p = 0; p < (int64_t)out.size(); ++p)
        if (out[p].front == sfront) where[key_of(out[p])] = p;
      auto find = [&](int type, int i, int j, int endp) -> int64_t {
        const auto it = where.find(((int64_t)type << 48) | ((int64_t)i << 32) | ((int64_t)j << 16) | (int64_t)endp);
        return it == where.end() ? -1 : it->second;
      };
      // task at position `from` to position anchor + 1 when that is earlier (everything in between moves one place back)
      auto hoist_to = [&](int64_t from, int64_t anchor) {
        const int64_t to = anchor + 1;
        if (from < 0 || from <= to) return;
        const DfTask t = out[from];
        for (int64_t p = from; p > to; --p) {
          out[p] = out[p - 1];
          if (out[p].front == sfront) where[key_of(out[p])] = p;
        }
        out[to] = t;
        where[key_of(t)] = to;
      };
      auto single = [&](int64_t p) { return p >= 0 && ((out[p].type_nq >> 8) & 255) == 1; };
      static const int near = getenv("OKKT_DF_HOIST_NEAR") ? atoi(getenv("OKKT_DF_HOIST_NEAR")) : 3;
      for (int q = 1; q + 1 < g.KB; ++q) {
        if (find(kDfT, q + 1, q - 1, q) < 0 || find(kDfTU, q, q - 1, q) < 0) continue;
        // the chain row's panel tile T(q + 1, q - 1) itself: `lead` places ahead of where the model dispatched it (it waits for D(q - 1),
        // which TU(q - 2) carries, and for its own last update) -- never in front of those
        static const int lead = getenv("OKKT_DF_HOIST_LEAD") ? atoi(getenv("OKKT_DF_HOIST_LEAD")) : 0;
        if (lead > 0) {
          const int64_t pt = find(kDfT, q + 1, q - 1, q);
          int64_t anchor = std::max(pt - 1 - lead, q >= 2 ? std::max(find(kDfTU, q - 1, q - 2, q - 1), find(kDfU, q + 1, q - 1, q - 1)) : find(kDfD, 0, 0, 1));
          if (q >= 2 && (find(kDfTU, q - 1, q - 2, q - 1) < 0 || find(kDfU, q + 1, q - 1, q - 1) < 0)) anchor = pt;      // unknown dependency position: stay
          anchor = std::max(anchor, find(kDfD, q - 1, q - 1, q));      // D(q - 1) as a task of its own
          hoist_to(pt, anchor);
        }
        // U(i, q; panel q - 1) right behind T(i, q - 1) for the chain row i = q + 1 and the rows next to it (they gate T(i, q), which
        // feeds the chain one and two steps later): needs that panel tile, the block row of TU(q - 1) and its own previous update
        for (int i = q + 1; i <= q + near && i < g.TB; ++i) {
          const int64_t pu = find(kDfU, i, q, q);
          if (!single(pu) || find(kDfT, i, q - 1, q) < 0) continue;
          if (q >= 2 && find(kDfU, i, q, q - 1) < 0) continue;
          const int64_t anchor = std::max(std::max(find(kDfT, i, q - 1, q), find(kDfTU, q, q - 1, q)), q >= 2 ? find(kDfU, i, q, q - 1) : (int64_t)-1);
          hoist_to(pu, anchor);
        }
        // U(q + 1, q + 1; panel q - 1): both operands are block row q + 1 of panel q - 1
        {
          const int64_t pu = find(kDfU, q + 1, q + 1, q);
          if (single(pu) && !(q >= 2 && find(kDfU, q + 1, q + 1, q - 1) < 0))
            hoist_to(pu, std::max(std::max(find(kDfT, q + 1, q - 1, q), find(kDfU, q + 1, q, q)), q >= 2 ? find(kDfU, q + 1, q + 1, q - 1) : (int64_t)-1));
        }
        // TA(q) and TU(q) behind both, behind TU(q - 1) (which carries D(q)) and behind D(q) where it is a task of its own
        {
          const int64_t pu1 = find(kDfU, q + 1, q, q), pu2 = find(kDfU, q + 1, q + 1, q);
          if (pu1 < 0 || pu2 < 0) continue;
          int64_t anchor = std::max(std::max(pu1, pu2), std::max(find(kDfTU, q, q - 1, q), find(kDfD, q, q, q + 1)));
          const int64_t pta = find(kDfTA, q + 1, q, q + 1);
          if (pta >= 0) { hoist_to(pta, anchor); anchor = std::max(anchor, find(kDfTA, q + 1, q, q + 1)); }
          hoist_to(find(kDfTU, q + 1, q, q + 1), anchor);
        }
      }
    }
  }
}

}  // namespace okkt
