// Host symbolic analysis: ordering -> elimination tree -> postorder -> column counts ->
// relaxed supernodes -> front layout, scatter maps, extend-add lists, level schedule.
// See symbolic.h for the role of this step relative to the reference.
#include "symbolic.h"

#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <algorithm>
#include <atomic>
#include <functional>
#include <cstring>
#include <numeric>
#include <thread>
#include <stdexcept>
#include <string>

namespace okkt {

uint64_t hash_pattern(int64_t n, const int64_t* colptr, const int64_t* rowval) {
  // A 64-bit digest of (n, colptr, rowval): an identifier of the sparsity pattern for logs and statistics (the decision to skip a
  // re-analysis is an exact comparison in api.cpp, not this).  One multiply-rotate step per 64-bit word -- the byte-wise FNV-1a of
  // rounds 1-4 was 20 ms at S-metric, in every candidate thread of the analysis.
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](uint64_t v) { h = (h ^ v) * 0x9E3779B97F4A7C15ull; h = (h << 29) | (h >> 35); };
  mix((uint64_t)n);
  for (int64_t j = 0; j <= n; ++j) mix((uint64_t)colptr[j]);
  const int64_t nnz = colptr[n] - colptr[0];
  for (int64_t p = 0; p < nnz; ++p) mix((uint64_t)rowval[p]);
  h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
  return h;
}

namespace {

inline int64_t trapezoid(int64_t f, int64_t k) { return f * k - k * (k - 1) / 2; }

}  // namespace

// forced_order (optional): the elimination order to use instead of computing one (opts.ordering is then only recorded).
// stats_only: stop behind the column counts -- S.nnzL, S.flops_exact and S.perm are valid, nothing else.
// order_out (optional): receives the order that was computed (before the postorder).
// [0, n) in contiguous pieces on up to `want` host threads (OKKT_ANALYZE_THREADS caps it; 1 = the calling thread alone).  fn(lo, hi, piece)
// must not throw across the boundary: exceptions are caught per piece and the first message is returned.  A thread that cannot be
// created (pid / thread limits) is replaced by a call on the calling thread.
template <class F>
static std::string parallel_pieces(int64_t n, int want, F fn) {
  int nt = std::max(1, want);
  if (const char* e = getenv("OKKT_ANALYZE_THREADS")) nt = std::min(nt, std::max(1, atoi(e)));
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw > 0) nt = std::min<int>(nt, (int)hw);
  nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, n));
  std::vector<std::string> err(nt);
  auto run = [&](int t) {
    const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
    try { fn(lo, hi, t); }
    catch (const std::exception& ex) { err[t] = ex.what(); }
    catch (...) { err[t] = "analysis piece failed"; }
  };
  std::vector<std::thread> th;
  std::vector<int> inline_pieces;
  for (int t = 1; t < nt; ++t) {
    try { th.emplace_back(run, t); } catch (...) { inline_pieces.push_back(t); }
  }
  run(0);
  for (int t : inline_pieces) run(t);
  for (auto& x : th) x.join();
  for (const std::string& e : err) if (!e.empty()) return e;
  return "";
}

static std::string analyze_one(int64_t n64, const int64_t* colptr, const int64_t* rowval,
                               int index_base, const SymbolicOptions& opts,
                               const int64_t* user_perm, Symbolic& S,
                               const std::vector<int>* forced_order = nullptr, bool stats_only = false,
                               const std::atomic<bool>* cancel = nullptr) {
  if (n64 < 0 || n64 > 0x7ffffff0) return "matrix order out of range";
  if (index_base != 0 && index_base != 1) return "index_base must be 0 or 1";
  const int n = (int)n64;
  S = Symbolic();
  S.n = n;
  const int64_t base = index_base;
  if (colptr[0] != base) return "colptr[0] does not equal index_base";
  const int64_t nnz_in = colptr[n] - base;
  S.nnz_in = nnz_in;
  for (int j = 0; j < n; ++j)
    if (colptr[j + 1] < colptr[j]) return "colptr is not non-decreasing";
  for (int64_t p = 0; p < nnz_in; ++p) {
    int64_t i = rowval[p] - base;
    if (i < 0 || i >= n) return "row index out of range";
  }
  S.pattern_hash = hash_pattern(n, colptr, rowval);

  // OKKT_DEBUG_ANALYZE: seconds per phase on stderr
  const bool dbg_time = getenv("OKKT_DEBUG_ANALYZE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_time) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "okkt: analyze %-28s %.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
    t_last = t;
  };
  // ---- strictly-lower pattern of the input, symmetrised, de-duplicated -> graph for ordering
  std::vector<int64_t> gp(n + 1, 0);
  std::vector<int> gi;
  {
    std::vector<int64_t> cnt(n, 0);
    int64_t nlow = 0;
    for (int j = 0; j < n; ++j)
      for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
        int i = (int)(rowval[p] - base);
        if (i >= j) ++nlow;
        if (i > j) { ++cnt[i]; ++cnt[j]; }
      }
    S.nnz_lower = nlow;
    for (int i = 0; i < n; ++i) gp[i + 1] = gp[i] + cnt[i];
    gi.resize(gp[n]);
    std::vector<int64_t> fill(gp.begin(), gp.end() - 1);
    for (int j = 0; j < n; ++j)
      for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
        int i = (int)(rowval[p] - base);
        if (i > j) { gi[fill[i]++] = j; gi[fill[j]++] = i; }
      }
    // sort + unique each list: the sorts on pieces of the row range (3.6 M entries at S-metric), the compaction behind them only when
    // some list had a duplicate
    std::vector<int64_t> np(n + 1, 0);
    int64_t out = 0;
    {
      std::vector<int> len(n, 0);
      const std::string perr = parallel_pieces(n, gi.size() >= 400000 ? 8 : 1, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
          std::sort(gi.begin() + gp[i], gi.begin() + gp[i + 1]);
          len[i] = (int)(std::unique(gi.begin() + gp[i], gi.begin() + gp[i + 1]) - (gi.begin() + gp[i]));
        }
      });
      if (!perr.empty()) return perr;
      for (int i = 0; i < n; ++i) {
        if (len[i] != gp[i + 1] - gp[i]) S.has_duplicates = true;
        if (out != gp[i]) std::copy(gi.begin() + gp[i], gi.begin() + gp[i] + len[i], gi.begin() + out);
        out += len[i];
        np[i + 1] = out;
      }
    }
    gi.resize(out);
    gp.swap(np);
  }
  // duplicates on the diagonal
  {
    for (int j = 0; j < n && !S.has_duplicates; ++j) {
      int ndiag = 0;
      for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p)
        if (rowval[p] - base == j) ++ndiag;
      if (ndiag > 1) S.has_duplicates = true;
    }
  }

  lap("graph");
  // ---- ordering
  std::vector<int> order;
  if (forced_order) {
    order = *forced_order;
    S.ordering_used = opts.ordering == 3 ? 0 : opts.ordering;
  } else if (opts.ordering == 0 || opts.ordering == 3) {
    amd_order(n, gp, gi, order, cancel);
    if (cancel && (int)order.size() != n) return "cancelled";
  } else if (opts.ordering == 5) {
    int top_sep = -1;
    ml_nd_order(n, gp, gi, opts.mlnd_leaf, opts.mlnd_trials, order, &top_sep);
    if ((int)order.size() != n) amd_order(n, gp, gi, order); else { S.ordering_used = 5; S.top_separator = top_sep; }
  } else if (opts.ordering == 4) {
    level_nd_order(n, gp, gi, opts.nd_leaf, order);
    S.ordering_used = 4;
  } else if (opts.ordering == 1) {
    order.resize(n);
    std::iota(order.begin(), order.end(), 0);
  } else if (opts.ordering == 2) {
    if (!user_perm) return "ordering=user but no permutation was supplied";
    order.resize(n);
    std::vector<char> seen(n, 0);
    for (int k = 0; k < n; ++k) {
      int64_t v = user_perm[k];
      if (v < 0 || v >= n || seen[v]) return "user permutation is not a permutation of 0..n-1";
      seen[v] = 1;
      order[k] = (int)v;
    }
  } else {
    return "unknown ordering option";
  }
  if ((int)order.size() != n) return "internal error: ordering length";
  {
    std::vector<char> seen(n, 0);
    for (int v : order) { if (v < 0 || v >= n || seen[v]) return "internal error: ordering is not a permutation"; seen[v] = 1; }
  }

  lap("ordering");
  // ---- elimination tree + column counts in the pre-postorder numbering
  std::vector<int> ip0(n);
  for (int k = 0; k < n; ++k) ip0[order[k]] = k;
  // row lists (cols < row) of the permuted pattern, row after row in the new numbering: the writes are sequential, the reads one list per
  // vertex (asked for a few vertices ahead) -- the by-entry scatter of rounds 1-4 was a cache miss per entry
  auto build_rows = [&](const std::vector<int>& ord, const std::vector<int>& ip, std::vector<int64_t>& rp, std::vector<int>& ri) {
    rp.assign(n + 1, 0);
    ri.resize(gp[n] / 2);
    int64_t out = 0;
    for (int r = 0; r < n; ++r) {
      if (r + 4 < n) __builtin_prefetch(&gi[gp[ord[r + 4]]]);
      const int i = ord[r];
      for (int64_t p = gp[i]; p < gp[i + 1]; ++p) { const int b = ip[gi[p]]; if (b < r) ri[out++] = b; }
      rp[r + 1] = out;
    }
    ri.resize(out);
  };
  auto etree_of = [&](const std::vector<int64_t>& rp, const std::vector<int>& ri, std::vector<int>& parent) {
    parent.assign(n, -1);
    std::vector<int> anc(n, -1);
    for (int i = 0; i < n; ++i)
      for (int64_t p = rp[i]; p < rp[i + 1]; ++p) {
        int k = ri[p];
        while (k != -1 && k < i) {  // climb with path compression
          int nxt = anc[k];
          anc[k] = i;
          if (nxt == -1) { parent[k] = i; break; }
          k = nxt;
        }
      }
  };
  // Column counts of L from the skeleton of A (Gilbert, Ng & Peyton 1994, as in Davis' cs_counts): every entry
  // A(i, j), i > j, is looked at once and decides with the first-descendant test whether j is a leaf of row i's
  // subtree; overlaps are subtracted at least common ancestors (disjoint-set forest).  O(nnz(A) alpha(n)) instead
  // of the O(nnz(L)) row-subtree walk (187 M steps at S-metric).  `post` must be a postorder of `parent`;
  // cp/ci = the strictly-lower column lists.
  auto column_counts = [&](const std::vector<int>& parent, const std::vector<int>& post, const std::vector<int64_t>& cp,
                           const std::vector<int>& ci, std::vector<int>& count) {
    std::vector<int> first(n, -1), maxfirst(n, -1), prevleaf(n, -1), ancestor(n);
    std::vector<int64_t> delta(n, 0);
    for (int k = 0; k < n; ++k) {
      int j = post[k];
      delta[j] = first[j] == -1 ? 1 : 0;                       // j is a leaf of the etree
      for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
    }
    for (int i = 0; i < n; ++i) ancestor[i] = i;
    for (int k = 0; k < n; ++k) {
      const int j = post[k];
      if (parent[j] != -1) --delta[parent[j]];                 // j is not a root
      for (int64_t p = cp[j]; p < cp[j + 1]; ++p) {
        const int i = ci[p];
        if (i <= j || first[j] <= maxfirst[i]) continue;       // j is not a leaf of the row subtree of i
        maxfirst[i] = first[j];
        const int jprev = prevleaf[i];
        prevleaf[i] = j;
        ++delta[j];                                            // A(i, j) is in the skeleton
        if (jprev != -1) {                                     // a later leaf: subtract the overlap at the lca
          int q = jprev;
          while (q != ancestor[q]) q = ancestor[q];
          for (int t = jprev; t != q;) { const int nxt = ancestor[t]; ancestor[t] = q; t = nxt; }
          --delta[q];
        }
      }
      if (parent[j] != -1) ancestor[j] = parent[j];
    }
    for (int k = 0; k < n; ++k) {                              // sum up the subtrees, children before parents
      const int j = post[k];
      if (parent[j] != -1) delta[parent[j]] += delta[j];
    }
    count.resize(n);
    for (int j = 0; j < n; ++j) count[j] = (int)delta[j];
  };
  auto plain_postorder = [&](const std::vector<int>& parent, std::vector<int>& post) {
    std::vector<int> head(n, -1), nxt(n, -1), stack;
    for (int j = n - 1; j >= 0; --j)
      if (parent[j] >= 0) { nxt[j] = head[parent[j]]; head[parent[j]] = j; }
    post.clear();
    post.reserve(n);
    for (int r = 0; r < n; ++r) {
      if (parent[r] >= 0) continue;
      stack.push_back(r);
      while (!stack.empty()) {
        const int v = stack.back();
        const int c = head[v];
        if (c >= 0) { head[v] = nxt[c]; stack.push_back(c); }
        else { post.push_back(v); stack.pop_back(); }
      }
    }
  };
  std::vector<int64_t> rp, cp;
  std::vector<int> ri, ci;
  std::vector<int> parent0, count0;
  build_rows(order, ip0, rp, ri);
  {
    // the tree (with a postorder of it) and the column lists both come from the row lists: side by side
    std::vector<int> post1;
    const std::string perr = parallel_pieces(2, n >= 20000 ? 2 : 1, [&](int64_t lo, int64_t hi, int) {
      for (int64_t piece = lo; piece < hi; ++piece) {
        if (piece == 0) { etree_of(rp, ri, parent0); plain_postorder(parent0, post1); }
        else {
          cp.assign(n + 1, 0);
          for (int64_t p = 0; p < rp[n]; ++p) ++cp[ri[p] + 1];
          for (int i = 0; i < n; ++i) cp[i + 1] += cp[i];
          ci.resize(rp[n]);
          std::vector<int64_t> f2(cp.begin(), cp.end() - 1);
          for (int r = 0; r < n; ++r)
            for (int64_t p = rp[r]; p < rp[r + 1]; ++p) ci[f2[ri[p]]++] = r;  // rows ascending per column
        }
      }
    });
    if (!perr.empty()) return perr;
    column_counts(parent0, post1, cp, ci, count0);
  }

  lap("etree + colcounts");
  // ---- postorder, heaviest child last (it is the amalgamation candidate of its parent)
  std::vector<int> post;
  post.reserve(n);
  {
    std::vector<int64_t> chp(n + 1, 0);
    for (int j = 0; j < n; ++j) if (parent0[j] >= 0) ++chp[parent0[j] + 1];
    for (int j = 0; j < n; ++j) chp[j + 1] += chp[j];
    std::vector<int> ch(chp[n]);
    std::vector<int64_t> fill(chp.begin(), chp.end() - 1);
    for (int j = 0; j < n; ++j) if (parent0[j] >= 0) ch[fill[parent0[j]]++] = j;
    for (int j = 0; j < n; ++j)
      std::stable_sort(ch.begin() + chp[j], ch.begin() + chp[j + 1],
                       [&](int a, int b) { return count0[a] < count0[b]; });
    std::vector<int> stack;
    std::vector<int64_t> next_child(n);
    for (int j = 0; j < n; ++j) next_child[j] = chp[j];
    std::vector<int> roots;
    for (int j = 0; j < n; ++j) if (parent0[j] < 0) roots.push_back(j);
    std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return count0[a] < count0[b]; });
    for (int r : roots) {
      stack.push_back(r);
      while (!stack.empty()) {
        int v = stack.back();
        if (next_child[v] < chp[v + 1]) stack.push_back(ch[next_child[v]++]);
        else { post.push_back(v); stack.pop_back(); }
      }
    }
  }
  S.perm.resize(n);
  S.iperm.resize(n);
  for (int k = 0; k < n; ++k) { S.perm[k] = order[post[k]]; S.iperm[S.perm[k]] = k; }
  // The postorder is a relabelling along the tree: the elimination tree and the column counts of the renumbered matrix are the
  // relabelled ones (no second pass over the pattern for them); only the column lists are built again, in the new numbering.
  {
    std::vector<int> inv_post(n);
    for (int k = 0; k < n; ++k) inv_post[post[k]] = k;
    S.parent.resize(n);
    S.colcount.resize(n);
    for (int k = 0; k < n; ++k) {
      const int j = post[k];
      S.parent[k] = parent0[j] >= 0 ? inv_post[parent0[j]] : -1;
      S.colcount[k] = count0[j];
    }
    // column lists with the rows ascending (the row-structure pass below reads them in that order): by new row number
    cp.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) {
      const int a = S.iperm[i];
      for (int64_t p = gp[i]; p < gp[i + 1]; ++p) { const int j = gi[p]; if (j < i) ++cp[std::min(a, S.iperm[j]) + 1]; }
    }
    for (int i = 0; i < n; ++i) cp[i + 1] += cp[i];
    ci.resize(cp[n]);
    std::vector<int64_t> f2(cp.begin(), cp.end() - 1);
    for (int r = 0; r < n; ++r) {
      const int i = S.perm[r];
      for (int64_t p = gp[i]; p < gp[i + 1]; ++p) { const int b = S.iperm[gi[p]]; if (b < r) ci[f2[b]++] = r; }
    }
  }
  const std::vector<int>& parent = S.parent;
  const std::vector<int>& cc = S.colcount;
  for (int j = 0; j < n; ++j) { S.nnzL += cc[j]; S.flops_exact += (double)cc[j] * cc[j]; }

  lap("postorder");
  if (stats_only) return "";
  // ---- fundamental supernodes, then relaxed amalgamation of (last child -> parent) chains
  std::vector<int> col0;  // first column of each supernode
  col0.reserve(n + 1);
  for (int j = 0; j < n; ++j) {
    bool join = j > 0 && parent[j - 1] == j && cc[j - 1] == cc[j] + 1;
    if (!join) col0.push_back(j);
  }
  col0.push_back(n);
  {
    int ns = (int)col0.size() - 1;
    std::vector<int> start(ns), width(ns);
    std::vector<int64_t> truennz(ns);
    std::vector<char> alive(ns, 1);
    for (int s = 0; s < ns; ++s) {
      start[s] = col0[s];
      width[s] = col0[s + 1] - col0[s];
      int64_t t = 0;
      for (int j = col0[s]; j < col0[s + 1]; ++j) t += cc[j];
      truennz[s] = t;
    }
    for (int s = 0; s + 1 < ns; ++s) {
      int last = col0[s + 1] - 1;
      if (parent[last] != last + 1) continue;  // parent supernode is not the next one
      int p = s + 1;
      int64_t kc = width[s], kp = width[p];
      int64_t fp = cc[col0[p]];  // front order of the parent: count of its first column
      int64_t fm = kc + fp, km = kc + kp;
      int64_t stored = trapezoid(fm, km);
      int64_t tn = truennz[s] + truennz[p];
      double z = stored > 0 ? (double)(stored - tn) / (double)stored : 0.0;
      bool merge = km <= opts.relax_always || (km <= opts.relax_small && z < opts.relax_small_frac) ||
                   (km <= opts.relax_mid && z < opts.relax_mid_frac) || z < opts.relax_any_frac;
      if (!merge) continue;
      alive[s] = 0;
      start[p] = start[s];
      width[p] = (int)km;
      truennz[p] = tn;
    }
    std::vector<int> merged;
    for (int s = 0; s < ns; ++s) if (alive[s]) merged.push_back(start[s]);
    merged.push_back(n);
    col0.swap(merged);
  }
  const int ns = (int)col0.size() - 1;
  S.nsuper = ns;
  S.sn_col0 = col0;
  S.col2sn.resize(n);
  for (int s = 0; s < ns; ++s)
    for (int j = col0[s]; j < col0[s + 1]; ++j) S.col2sn[j] = s;
  S.sn_parent.assign(ns, -1);
  for (int s = 0; s < ns; ++s) {
    int pj = parent[col0[s + 1] - 1];
    S.sn_parent[s] = pj >= 0 ? S.col2sn[pj] : -1;
  }
  S.child_ptr.assign(ns + 1, 0);
  for (int s = 0; s < ns; ++s) if (S.sn_parent[s] >= 0) ++S.child_ptr[S.sn_parent[s] + 1];
  for (int s = 0; s < ns; ++s) S.child_ptr[s + 1] += S.child_ptr[s];
  S.children.resize(S.child_ptr[ns]);
  {
    std::vector<int64_t> fill(S.child_ptr.begin(), S.child_ptr.end() - 1);
    for (int s = 0; s < ns; ++s) if (S.sn_parent[s] >= 0) S.children[fill[S.sn_parent[s]]++] = s;
  }

  lap("supernodes");
  // ---- row structure of every front (sorted union of A's columns and the children's rows)
  S.row_ptr.assign(ns + 1, 0);
  S.rows.clear();
  S.rows.reserve((size_t)S.nnzL / 2 + n);
  {
    std::vector<int> mark(n, -1);
    std::vector<int> extra;
    for (int s = 0; s < ns; ++s) {
      const int c0 = col0[s], c1 = col0[s + 1], last = c1 - 1;
      extra.clear();
      for (int j = c0; j < c1; ++j)
        for (int64_t p = cp[j]; p < cp[j + 1]; ++p) {
          int i = ci[p];
          if (i > last && mark[i] != s) { mark[i] = s; extra.push_back(i); }
        }
      for (int64_t q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) {
        int c = S.children[q];
        int kc = col0[c + 1] - col0[c];
        for (int64_t p = S.row_ptr[c] + kc; p < S.row_ptr[c + 1]; ++p) {
          int i = S.rows[p];
          if (i > last && mark[i] != s) { mark[i] = s; extra.push_back(i); }
        }
      }
      std::sort(extra.begin(), extra.end());
      for (int j = c0; j < c1; ++j) S.rows.push_back(j);
      S.rows.insert(S.rows.end(), extra.begin(), extra.end());
      S.row_ptr[s + 1] = (int64_t)S.rows.size();
      if ((int)extra.size() != cc[last] - 1) return "internal error: supernode row structure mismatch";
    }
  }

  lap("row structure");
  // ---- front layout, relative indices, solve workspaces, statistics
  S.front_pos.assign(ns + 1, 0);
  S.rel_ptr.assign(ns + 1, 0);
  S.cv_pos.assign(ns + 1, 0);
  for (int s = 0; s < ns; ++s) {
    int64_t f = S.row_ptr[s + 1] - S.row_ptr[s];
    int64_t k = col0[s + 1] - col0[s];
    int64_t r = f - k;
    int64_t sz = (f * f + 1) & ~(int64_t)1;  // keep every front 16-byte aligned
    S.front_pos[s + 1] = S.front_pos[s] + sz;
    S.rel_ptr[s + 1] = S.rel_ptr[s] + r;
    S.cv_pos[s + 1] = S.cv_pos[s] + r;
    S.nnzL_stored += trapezoid(f, k);
    for (int64_t j = 0; j < k; ++j) S.flops_stored += (double)(f - j) * (double)(f - j);
    S.max_front = std::max<int>(S.max_front, (int)f);
    S.sum_r += r;
  }
  S.arena_doubles = S.front_pos[ns];
  S.rel.resize(S.rel_ptr[ns]);
  {
    // every supernode's list on its own: pieces of the supernode range on host threads
    const std::string perr = parallel_pieces(ns, S.rel_ptr[ns] >= 200000 ? 8 : 1, [&](int64_t slo, int64_t shi, int) {
      for (int64_t s = slo; s < shi; ++s) {
        int p = S.sn_parent[s];
        if (p < 0) continue;
        int k = col0[s + 1] - col0[s];
        const int* mine = &S.rows[S.row_ptr[s] + k];
        int64_t r = S.row_ptr[s + 1] - S.row_ptr[s] - k;
        const int* theirs = &S.rows[S.row_ptr[p]];
        int64_t fp = S.row_ptr[p + 1] - S.row_ptr[p];
        int64_t q = 0;
        for (int64_t i = 0; i < r; ++i) {
          while (q < fp && theirs[q] < mine[i]) ++q;
          if (q >= fp || theirs[q] != mine[i]) throw std::runtime_error("internal error: child row missing from parent front");
          S.rel[S.rel_ptr[s] + i] = (int)q;
        }
      }
    });
    if (!perr.empty()) return perr;
  }

  lap("layout");
  // ---- scatter maps from the caller's nzval into the front arena
  S.amap.assign(nnz_in, -1);
  S.diag_pos.assign(n, -1);
  for (int j = 0; j < n; ++j) {
    int c = S.iperm[j];
    int s = S.col2sn[c];
    int64_t f = S.row_ptr[s + 1] - S.row_ptr[s];
    int64_t lc = c - col0[s];
    S.diag_pos[j] = S.front_pos[s] + lc * f + lc;
  }
  {
    std::vector<int64_t> cnt(ns + 1, 0);
    std::vector<int> ent_sn((size_t)nnz_in, -1);     // destination supernode of every used entry
    // position of a row inside the front of supernode s: the pivot rows are the columns themselves (no search), the rest
    // by binary search in the sorted tail of the row list.  Columns are independent: pieces of the column range on host threads
    // (3.6 M binary searches at S-metric: 0.06 s on one thread).
    const int map_threads = nnz_in >= 200000 ? 8 : 1;
    {
      const std::string perr = parallel_pieces(n, map_threads, [&](int64_t jlo, int64_t jhi, int) {
        for (int64_t j = jlo; j < jhi; ++j)
          for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
            int i = (int)(rowval[p] - base);
            if (i < j) continue;  // upper-triangle entries are ignored (Symmetric(A,:L) in the reference)
            int a = S.iperm[i], b = S.iperm[j];
            int c = std::min(a, b), r = std::max(a, b);
            int s = S.col2sn[c];
            const int* rb = &S.rows[S.row_ptr[s]];
            const int* re = &S.rows[S.row_ptr[s + 1]];
            const int ksn = col0[s + 1] - col0[s];
            int64_t pos;
            if (r < col0[s + 1]) pos = r - col0[s];
            else {
              const int* it = std::lower_bound(rb + ksn, re, r);
              if (it == re || *it != r) throw std::runtime_error("internal error: entry outside the symbolic structure");
              pos = it - rb;
            }
            int64_t f = re - rb;
            int64_t off = (int64_t)(c - col0[s]) * f + pos;
            S.amap[p] = S.front_pos[s] + off;
            ent_sn[p] = s;
          }
      });
      if (!perr.empty()) return perr;
    }
    // counting sort of the entries by supernode, the entries of one supernode in input order: pieces of the entry range count on
    // their own, a piece writes behind the pieces before it (the same lists for every piece count)
    {
      const int np = map_threads;
      std::vector<std::vector<int>> pcnt(np);
      std::vector<int64_t> plo(np, 0), phi(np, 0);
      std::string perr = parallel_pieces(nnz_in, np, [&](int64_t lo, int64_t hi, int t) {
        plo[t] = lo; phi[t] = hi;
        std::vector<int>& c = pcnt[t];
        c.assign(ns, 0);
        for (int64_t p = lo; p < hi; ++p) if (ent_sn[p] >= 0) ++c[ent_sn[p]];
      });
      if (!perr.empty()) return perr;
      S.aent_ptr.assign(ns + 1, 0);
      for (int s = 0; s < ns; ++s) {
        int64_t c = 0;
        for (int t = 0; t < np; ++t) if (!pcnt[t].empty()) { const int x = pcnt[t][s]; pcnt[t][s] = (int)c; c += x; }   // offset of piece t inside supernode s
        cnt[s + 1] = c;
        S.aent_ptr[s + 1] = S.aent_ptr[s] + c;
      }
      S.aent_src.resize(S.aent_ptr[ns]);
      S.aent_dst.resize(S.aent_ptr[ns]);
      perr = parallel_pieces(np, np, [&](int64_t tlo, int64_t thi, int) {
        for (int64_t t = tlo; t < thi; ++t) {
          if (pcnt[t].empty()) continue;
          std::vector<int>& off = pcnt[t];
          for (int64_t p = plo[t]; p < phi[t]; ++p) {
            const int s = ent_sn[p];
            if (s < 0) continue;
            const int64_t q = S.aent_ptr[s] + off[s]++;
            S.aent_src[q] = p;
            S.aent_dst[q] = (int)(S.amap[p] - S.front_pos[s]);
          }
        }
      });
      if (!perr.empty()) return perr;
    }
    // inside a supernode keep the entries sorted by destination: the big-front assemble kernel
    // locates the entries of a column block by binary search
    {
      const std::string perr = parallel_pieces(ns, map_threads, [&](int64_t slo, int64_t shi, int) {
        std::vector<std::pair<int, int64_t>> tmp;
        for (int64_t s = slo; s < shi; ++s) {
          const int64_t e0 = S.aent_ptr[s], e1 = S.aent_ptr[s + 1];
          bool sorted = true;
          for (int64_t e = e0 + 1; e < e1; ++e) if (S.aent_dst[e - 1] > S.aent_dst[e]) { sorted = false; break; }
          if (sorted) continue;
          tmp.clear();
          for (int64_t e = e0; e < e1; ++e) tmp.emplace_back(S.aent_dst[e], S.aent_src[e]);
          std::stable_sort(tmp.begin(), tmp.end());
          for (int64_t e = e0; e < e1; ++e) { S.aent_dst[e] = tmp[e - e0].first; S.aent_src[e] = tmp[e - e0].second; }
        }
      });
      if (!perr.empty()) return perr;
    }
  }

  lap("scatter maps");
  if (dbg_time) {
    long hist[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int sn = 0; sn < (int)S.sn_col0.size() - 1; ++sn) {
      const int64_t f = S.row_ptr[sn + 1] - S.row_ptr[sn];
      ++hist[f <= 8 ? 0 : f <= 16 ? 1 : f <= 24 ? 2 : f <= 32 ? 3 : f <= 64 ? 4 : f <= 128 ? 5 : 6];
    }
    fprintf(stderr, "okkt: analyze fronts by order: <=8 %ld  <=16 %ld  <=24 %ld  <=32 %ld  <=64 %ld  <=128 %ld  larger %ld\n", hist[0], hist[1],
            hist[2], hist[3], hist[4], hist[5], hist[6]);
  }
  // ---- level schedule (height above the leaves)
  S.sn_level.assign(ns, 0);
  for (int s = 0; s < ns; ++s) {
    int p = S.sn_parent[s];
    if (p >= 0) S.sn_level[p] = std::max(S.sn_level[p], S.sn_level[s] + 1);
  }
  S.nlevels = 0;
  for (int s = 0; s < ns; ++s) S.nlevels = std::max(S.nlevels, S.sn_level[s] + 1);
  S.level_ptr.assign(S.nlevels + 1, 0);
  for (int s = 0; s < ns; ++s) ++S.level_ptr[S.sn_level[s] + 1];
  for (int l = 0; l < S.nlevels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
  S.level_sn.resize(ns);
  {
    std::vector<int> fill(S.level_ptr.begin(), S.level_ptr.end() - 1);
    for (int s = 0; s < ns; ++s) S.level_sn[fill[S.sn_level[s]]++] = s;
  }
  // pivots on the longest leaf-to-root path: the length of the dependency chain a factorisation cannot shorten
  {
    std::vector<int64_t> cp(ns, 0);
    S.critical_pivots = 0;
    for (int s = 0; s < ns; ++s) {          // postorder: children before parents
      cp[s] += S.sn_col0[s + 1] - S.sn_col0[s];
      S.critical_pivots = std::max(S.critical_pivots, cp[s]);
      const int p = S.sn_parent[s];
      if (p >= 0) cp[p] = std::max(cp[p], cp[s]);
    }
  }
  if (opts.ordering == 1 || opts.ordering == 2) S.ordering_used = opts.ordering;
  lap("levels, chain");
  return "";
}

// ordering = 0 (automatic):
//  * systems with real arithmetic in them (n >= 10 000) get two candidate orderings, computed side by side on two host
//    threads: approximate minimum degree (what the reference's CHOLMOD call uses) and multilevel nested dissection
//    (mlnd.cpp).  The one with fewer factor flops (sum of squared column counts) is analysed in full.  S-metric: 1.78e12
//    (AMD) against 5.7e11 (dissection), S-C3 3.9e10 against 1.9e10.
//  * AMD first otherwise.  When its elimination tree is (close to) a path of small fronts -- a banded KKT system, BASELINE
//    config 2 -- the factorisation would be one dependent pivot after the other: the analysis is redone with level-structure
//    nested dissection (nd.cpp) and that plan is kept when its dependency chain is at least three times shorter and the extra
//    fill stays small in absolute terms (such systems carry a few MFLOP; the chain length, not the flop count, is their cost).
std::string analyze_pattern(int64_t n64, const int64_t* colptr, const int64_t* rowval,
                            int index_base, const SymbolicOptions& opts,
                            const int64_t* user_perm, Symbolic& S) {
  const bool dbg = getenv("OKKT_DEBUG_ANALYZE") != nullptr;
  const int64_t mlnd_min_n = getenv("OKKT_MLND_MIN_N") ? atoll(getenv("OKKT_MLND_MIN_N")) : 10000;
  if (opts.ordering == 0 && n64 >= mlnd_min_n && n64 <= 0x7ffffff0) {
    SymbolicOptions oa = opts, ob = opts, oc = opts;
    oa.ordering = 3;
    ob.ordering = 5;
    oc.ordering = 4;
    Symbolic Sa, Sb, Sc;
    std::string ea, eb, ec, espec = "not run";
    // third candidate (round 3): the level-structure dissection.  On mesh-like graphs (discretised PDE constraints) its separators
    // are the grid planes: 40^3 grid 1.0e10 flops against 2.1e10 (multilevel dissection) and 4.6e10 (minimum degree), 400^2 grid
    // 6.1e8 against 1.2e9 and 1.0e9; on graphs with long-range edges it is hopeless (S-metric: 50 x more flops) and loses the
    // comparison.  It costs 0.1 s of one more thread, statistics only.
    // The candidate threads never let an exception escape (a std::system_error from thread creation under a pid / thread limit, a
    // bad_alloc inside a thread would otherwise end the host process through std::terminate): a candidate that fails is simply not a
    // candidate, minimum degree on the calling thread is the fall-back, and the joins run whatever happens (advisor, round 3).
    // OKKT_ANALYZE_THREADS=1 runs the candidates one after the other on the calling thread.
    auto run_c = [&] {
      try { ec = analyze_one(n64, colptr, rowval, index_base, oc, user_perm, Sc, nullptr, true); }
      catch (const std::exception& ex) { ec = std::string("level-structure candidate: ") + ex.what(); }
      catch (...) { ec = "level-structure candidate failed"; }
    };
    // The dissection is the faster of the two on many-core hosts (its pieces are ordered in parallel, minimum degree is one
    // thread): its thread goes on with the FULL analysis of its own ordering while minimum degree is still running -- the plan
    // is ready when the comparison is decided, and is thrown away when minimum degree wins
    // (round 4: ONE pass -- the candidate's statistics are the first half of the full analysis; the second pass over the graph, the
    // elimination tree, the column counts and the postorder cost 0.05 s of the 0.6)
    // (round 5) Minimum degree (0.47 - 0.51 s on one thread at S-metric) was the critical path of the analysis only to be COMPARED with:
    // when the dissection's plan is ready and its top separator is small -- at most trust_frac of the graph: the graph has the
    // separators the method lives on (S-metric 7 %, S-C5 0.2 %; the fully random S-C3 variant, where minimum degree wins: above 25 %) --
    // the minimum-degree candidate is abandoned where it stands and the dissection is taken without a flop comparison.
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    static const double trust_frac = getenv("OKKT_MLND_TRUST_FRAC") ? atof(getenv("OKKT_MLND_TRUST_FRAC")) : 0.12;
    std::atomic<bool> cancel_amd{false};
    auto run_b = [&] {
      try {
        eb = analyze_one(n64, colptr, rowval, index_base, ob, user_perm, Sb, nullptr, false);
        espec = eb.empty() && Sb.ordering_used == 5 ? "" : "no dissection plan";
        if (espec.empty() && Sb.top_separator >= 0 && (double)Sb.top_separator <= trust_frac * (double)n64 && Sb.flops_exact >= 1e9) cancel_amd.store(true);
        if (dbg) fprintf(stderr, "okkt: analyze dissection plan ready at %.3f s\n", since());
      } catch (const std::exception& ex) { eb = std::string("dissection candidate: ") + ex.what(); espec = eb; }
      catch (...) { eb = "dissection candidate failed"; espec = eb; }
    };
    const bool serial = getenv("OKKT_ANALYZE_THREADS") && atoi(getenv("OKKT_ANALYZE_THREADS")) <= 1;
    std::thread tb, tc;
    bool b_started = false, c_started = false;
    if (!serial) {
      try { tc = std::thread(run_c); c_started = true; } catch (...) { c_started = false; }
      try { tb = std::thread(run_b); b_started = true; } catch (...) { b_started = false; }
    }
    // (serial mode: the dissection first, so that minimum degree can be skipped altogether)
    if (!b_started) run_b();
    try { ea = analyze_one(n64, colptr, rowval, index_base, oa, user_perm, Sa, nullptr, true, &cancel_amd); }
    catch (const std::exception& ex) { ea = std::string("minimum-degree analysis: ") + ex.what(); }
    catch (...) { ea = "minimum-degree analysis failed"; }
    const auto tj0 = std::chrono::steady_clock::now();
    if (dbg) fprintf(stderr, "okkt: analyze minimum degree returned at %.3f s (%s)\n", since(), ea.c_str());
    if (b_started) tb.join();
    const auto tj1 = std::chrono::steady_clock::now();
    if (c_started) tc.join(); else run_c();
    if (dbg) fprintf(stderr, "okkt: analyze joins: minimum degree returned, then %.3f s for the dissection thread, %.3f s for the level-structure thread\n",
                     std::chrono::duration<double>(tj1 - tj0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tj1).count());
    // The decision is a function of the DATA only (advisor, round 5): the trust rule is evaluated again here, behind both joins, whether
    // or not minimum degree happened to finish before the dissection thread raised the flag -- the flag only saves time.  Every rank of a
    // sharded run and every repetition on a loaded host therefore takes the same permutation.
    const bool trusted = espec.empty() && Sb.top_separator >= 0 && (double)Sb.top_separator <= trust_frac * (double)n64 && Sb.flops_exact >= 1e9;
    const bool amd_done = ea.empty();      // minimum degree ran to the end: its statistics are reported (flops_other), they do not decide
    if (trusted) {
      // the level-structure candidate still has a say (mesh-like graphs): it must beat the dissection by 10 %
      const bool lv = ec.empty() && Sc.ordering_used == 4 && Sc.flops_exact < 0.9 * Sb.flops_exact;
      if (dbg)
        fprintf(stderr, "okkt: analyze candidates: AMD %s (top separator %ld of %ld) | nested dissection flops %.4g nnz(L) %ld | level-structure dissection flops %.4g -> %s\n",
                amd_done ? "not consulted" : "abandoned", (long)Sb.top_separator, (long)n64, Sb.flops_exact, (long)Sb.nnzL, Sc.flops_exact, lv ? "level-structure dissection" : "nested dissection");
      if (lv) {
        const std::vector<int> ord = Sc.perm;
        std::string e2 = analyze_one(n64, colptr, rowval, index_base, oc, user_perm, S, &ord, false);
        S.amd_skipped = !amd_done; S.flops_other = Sb.flops_exact;
        return e2;
      }
      const double fa = amd_done ? Sa.flops_exact : 0.0;
      S = std::move(Sb);
      S.amd_skipped = !amd_done;
      if (amd_done) S.flops_other = fa;      // the loser's flops when they are known anyway (advisor: a silently worse plan shows in the stats)
      if (dbg) fprintf(stderr, "okkt: analyze plan taken at %.3f s\n", since());
      return "";
    }
    if (!ea.empty()) return ea;
    const bool b_ok = eb.empty() && Sb.ordering_used == 5 && Sb.flops_exact < 0.9 * Sa.flops_exact && Sa.flops_exact >= 1e9;
    const bool c_ok = ec.empty() && Sc.ordering_used == 4 && Sc.flops_exact < 0.9 * Sa.flops_exact && Sa.flops_exact >= 1e9;
    const bool lv_wins = c_ok && (!b_ok || Sc.flops_exact < 0.9 * Sb.flops_exact);      // the multilevel plan is already built: the level structure has to be clearly better
    const bool nd_wins = b_ok && !lv_wins;
    if (dbg)
      fprintf(stderr, "okkt: analyze candidates: AMD flops %.4g nnz(L) %ld | nested dissection flops %.4g nnz(L) %ld | level-structure dissection flops %.4g nnz(L) %ld -> %s\n",
              Sa.flops_exact, (long)Sa.nnzL, Sb.flops_exact, (long)Sb.nnzL, Sc.flops_exact, (long)Sc.nnzL, lv_wins ? "level-structure dissection" : (nd_wins ? "nested dissection" : "AMD"));
    if (lv_wins) {
      const std::vector<int> ord = Sc.perm;
      return analyze_one(n64, colptr, rowval, index_base, oc, user_perm, S, &ord, false);
    }
    if (nd_wins) {
      if (espec.empty()) { const double fa = Sa.flops_exact; S = std::move(Sb); S.flops_other = fa; return ""; }
      const std::vector<int> ord = Sb.perm;
      return analyze_one(n64, colptr, rowval, index_base, ob, user_perm, S, &ord, false);
    }
    // AMD: fall through to the path-like rule with the order already known
    const std::vector<int> ord = Sa.perm;
    std::string e = analyze_one(n64, colptr, rowval, index_base, oa, user_perm, S, &ord, false);
    if (!e.empty()) return e;
    if (eb.empty() && Sb.ordering_used == 5) S.flops_other = Sb.flops_exact;
  } else {
    std::string e = analyze_one(n64, colptr, rowval, index_base, opts, user_perm, S);
    if (!e.empty() || opts.ordering != 0) return e;
  }
  std::string e;
  const bool path_like = S.n >= 128 && S.max_front <= 256 && S.critical_pivots * 5 >= S.n;
  if (!path_like) return e;
  SymbolicOptions o2 = opts;
  o2.ordering = 4;
  // no amalgamation beyond 4 columns: merging a child into its parent puts the child's pivots on the parent's level and
  // lengthens the dependency chain (S-C2, N_h = 20000: 0.37 ms with width 4, 0.56 with 16, 0.60 with 64; AMD path: 70.8 ms)
  o2.relax_always = o2.relax_small = o2.relax_mid = 4;
  if (getenv("OKKT_ND_LEAF")) o2.nd_leaf = atoi(getenv("OKKT_ND_LEAF"));
  if (getenv("OKKT_ND_RELAX")) o2.relax_always = o2.relax_small = o2.relax_mid = atoi(getenv("OKKT_ND_RELAX"));
  Symbolic S2;
  if (!analyze_one(n64, colptr, rowval, index_base, o2, user_perm, S2).empty()) return e;
  if (dbg)
    fprintf(stderr, "okkt: analyze path-like tree: AMD chain %ld of %ld pivots, flops %.3g, max front %d; nested dissection chain %ld, flops %.3g, max front %d\n",
            (long)S.critical_pivots, (long)S.n, S.flops_stored, S.max_front, (long)S2.critical_pivots, S2.flops_stored, S2.max_front);
  if (S2.critical_pivots * 3 <= S.critical_pivots && S2.flops_stored <= 20.0 * S.flops_stored + 1e8 && S2.max_front <= 512) S = std::move(S2);
  return e;
}


void partition_tree(Symbolic& S, int nparts) {
  const int ns = S.nsuper;
  S.nparts = std::max(1, nparts);
  S.sn_owner.assign(ns, 0);
  S.boundary.clear();
  S.boundary_cb.assign(1, 0);
  S.boundary_cv.assign(1, 0);
  S.part_flops.assign(S.nparts, 0.0);
  S.top_flops = 0;
  std::vector<double> own(ns, 0.0), sub(ns, 0.0);
  for (int s = 0; s < ns; ++s) {
    const double f = (double)(S.row_ptr[s + 1] - S.row_ptr[s]);
    const int k = S.sn_col0[s + 1] - S.sn_col0[s];
    for (int j = 0; j < k; ++j) own[s] += (f - j) * (f - j);
  }
  for (int s = 0; s < ns; ++s) {   // children precede parents
    sub[s] += own[s];
    if (S.sn_parent[s] >= 0) sub[S.sn_parent[s]] += sub[s];
  }
  if (S.nparts == 1) {
    for (int s = 0; s < ns; ++s) S.part_flops[0] += own[s];
    return;
  }
  // candidates = current subtree roots; top = nodes that were split.  Greedy on the estimated parallel
  // time  T = (flops of the top, serial on part 0) + (heaviest bin of an LPT packing of the candidates):
  // the heaviest candidate is split into its children while that lowers T.
  std::vector<char> is_top(ns, 0);
  std::vector<int> cand;
  for (int s = 0; s < ns; ++s) if (S.sn_parent[s] < 0) cand.push_back(s);
  auto lpt_max = [&](const std::vector<int>& cs) {
    std::vector<double> w;
    w.reserve(cs.size());
    for (int s : cs) w.push_back(sub[s]);
    std::sort(w.begin(), w.end(), std::greater<double>());
    std::vector<double> bins(S.nparts, 0.0);
    for (double x : w) *std::min_element(bins.begin(), bins.end()) += x;
    return *std::max_element(bins.begin(), bins.end());
  };
  double top = 0.0;
  double best_T = top + lpt_max(cand);
  // The walk may pass through non-improving splits (a chain of single-child fronts -- the dense top of a block -- has to
  // be crossed before the tree widens); the state with the best T seen is what is kept: everything split after it is
  // rolled back.  (Round 1 kept the final state of the walk: on the block-angular S-C5 that moved the dense roots of all
  // eight blocks into the serial top, 25 % of the flops instead of 7 %, estimated speed-up 2.9 instead of 4.8 at 8 parts.)
  std::vector<int> best_cand = cand, split_order;
  size_t best_nsplit = 0;
  double best_top = 0.0;
  const int max_stall = 64;
  int stall = 0;
  for (int iter = 0; iter < ns && stall < max_stall; ++iter) {
    int best = -1;
    for (size_t q = 0; q < cand.size(); ++q) {
      const bool hc = S.child_ptr[cand[q] + 1] > S.child_ptr[cand[q]];
      if (!hc) continue;
      if (best < 0 || sub[cand[q]] > sub[cand[best]] || (sub[cand[q]] == sub[cand[best]] && cand[q] < cand[best])) best = (int)q;
    }
    if (best < 0) break;
    const int s = cand[best];
    cand[best] = cand.back();
    cand.pop_back();
    for (int64_t q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) cand.push_back(S.children[q]);
    top += own[s];
    split_order.push_back(s);
    const double T = top + lpt_max(cand);
    if (T < best_T * (1.0 - 1e-12)) {
      best_T = T; stall = 0;
      best_cand = cand; best_nsplit = split_order.size(); best_top = top;
    } else {
      ++stall;
    }
  }
  cand = best_cand;
  top = best_top;
  for (size_t q = 0; q < best_nsplit; ++q) is_top[split_order[q]] = 1;
  S.top_flops = top;
  // LPT bin packing of the candidate subtrees
  std::sort(cand.begin(), cand.end(), [&](int a, int b) { return sub[a] > sub[b] || (sub[a] == sub[b] && a < b); });
  std::vector<int> root_owner(ns, -2);
  for (int s : cand) {
    int p = 0;
    for (int q = 1; q < S.nparts; ++q) if (S.part_flops[q] < S.part_flops[p]) p = q;
    root_owner[s] = p;
    S.part_flops[p] += sub[s];
  }
  // propagate ownership down the subtrees (parents have larger indices: walk downwards)
  for (int s = ns - 1; s >= 0; --s) {
    if (is_top[s]) { S.sn_owner[s] = -1; continue; }
    if (root_owner[s] >= 0) { S.sn_owner[s] = root_owner[s]; continue; }
    S.sn_owner[s] = S.sn_owner[S.sn_parent[s]];
  }
  for (int s = 0; s < ns; ++s) {
    const int p = S.sn_parent[s];
    if (S.sn_owner[s] >= 0 && p >= 0 && S.sn_owner[p] == -1) {
      const int64_t k = S.sn_col0[s + 1] - S.sn_col0[s];
      const int64_t r = S.row_ptr[s + 1] - S.row_ptr[s] - k;
      S.boundary.push_back(s);
      S.boundary_cb.push_back(S.boundary_cb.back() + r * r);
      S.boundary_cv.push_back(S.boundary_cv.back() + r);
    }
  }
}

}  // namespace okkt
