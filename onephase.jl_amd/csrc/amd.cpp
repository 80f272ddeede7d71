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
//
// Round 2: the quotient graph lives in ONE integer workspace (every variable / element owns a contiguous list: its elements
// first, then its variables; new elements are appended, dead space is reclaimed by an in-place compaction) instead of
// three std::vector per node -- the ordering was 75 % of the analysis (0.6 s of 0.8 s at n + m = 1e5) and most of that was
// allocator traffic and pointer chasing.  Same algorithm: approximate external degrees, aggressive element absorption, mass
// elimination, supervariable detection by hashing, dense rows ordered last.
#include "symbolic.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace okkt {

namespace {

enum : uint8_t { ST_VAR = 0, ST_ELEM = 1, ST_DEAD = 2, ST_ABSORBED = 3, ST_DENSE = 4 };

}  // namespace

void amd_order(int n, const std::vector<int64_t>& ap, const std::vector<int>& ai,
               std::vector<int>& order, const std::atomic<bool>* cancel) {
  order.clear();
  order.reserve(n);
  if (n == 0) return;

  // one 32-byte record per node: the scans touch pe, len, elen, nv, degree and state of every member of a new element, and
  // six separate arrays cost six cache misses per member (the scans are pure memory latency)
  struct Node { int64_t pe = -1; int len = 0, elen = 0, nv = 1, degree = 0; uint8_t state = ST_VAR; };
  std::vector<Node> nd(n);
  // dense rows are taken out and ordered last (they would join every element)
  const int dense_thr = std::max(16, (int)(10.0 * std::sqrt((double)n)));
  std::vector<int> dense_nodes;
  for (int i = 0; i < n; ++i)
    if (ap[i + 1] - ap[i] > dense_thr) { nd[i].state = ST_DENSE; dense_nodes.push_back(i); }
  const int nleft = n - (int)dense_nodes.size();

  // ---- workspace: nd[i].pe start of i's list, nd[i].len its length, nd[i].elen how many of its leading entries are elements
  int64_t nz = 0;
  for (int i = 0; i < n; ++i) {
    if (nd[i].state == ST_DENSE) continue;
    for (int64_t p = ap[i]; p < ap[i + 1]; ++p) { const int j = ai[p]; if (j != i && nd[j].state != ST_DENSE) ++nz; }
  }
  std::vector<int> iw((size_t)(nz + nz / 4 + 2 * (int64_t)n + 64));
  int64_t pfree = 0;
  for (int i = 0; i < n; ++i) {
    if (nd[i].state == ST_DENSE) { nd[i].nv = 0; continue; }
    nd[i].pe = pfree;
    for (int64_t p = ap[i]; p < ap[i + 1]; ++p) { const int j = ai[p]; if (j != i && nd[j].state != ST_DENSE) iw[pfree++] = j; }
    nd[i].len = (int)(pfree - nd[i].pe);
    nd[i].degree = nd[i].len;
  }
  // degree lists
  std::vector<int> head(n + 1, -1), next(n, -1), prev(n, -1);
  int mindeg = n;
  auto dl_insert = [&](int i, int d) {
    next[i] = head[d]; prev[i] = -1;
    if (head[d] >= 0) prev[head[d]] = i;
    head[d] = i;
    if (d < mindeg) mindeg = d;
  };
  auto dl_remove = [&](int i, int d) {
    if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d] = next[i];
    if (next[i] >= 0) prev[next[i]] = prev[i];
  };
  for (int i = n - 1; i >= 0; --i) if (nd[i].state == ST_VAR) dl_insert(i, nd[i].degree);

  std::vector<int64_t> w(n, 0);      // w[e] - wflg = |L_e \\ L_me| while an element is being compared; 0 = dead element
  for (int i = 0; i < n; ++i) w[i] = 1;
  int64_t wflg = 2;
  std::vector<int> hhead(n, -1), hnext(n, -1), absorbed_into(n, -1), touched, keep;
  std::vector<int64_t> stamp(n, -1);
  int64_t stamp_tag = 0;
  std::vector<int> pivots;
  pivots.reserve(n);

  // in-place compaction of the workspace: every live list moves to the front, order of the lists preserved
  auto compact = [&]() {
    for (int i = 0; i < n; ++i)
      if (nd[i].pe >= 0 && nd[i].len > 0) { const int64_t p = nd[i].pe; nd[i].pe = iw[p]; iw[p] = -(i + 1); }   // first entry parked in pe, marker in its place
      else if (nd[i].pe >= 0) nd[i].pe = -1;                                                            // an empty live list needs no space
    int64_t src = 0, dst = 0;
    while (src < pfree) {
      const int v = iw[src++];
      if (v >= 0) continue;                      // dead space
      const int i = -v - 1;
      iw[dst] = (int)nd[i].pe;
      nd[i].pe = dst++;
      for (int q = 1; q < nd[i].len; ++q) iw[dst++] = iw[src++];
    }
    pfree = dst;
  };

  int eliminated = 0;
  const bool dbg = getenv("OKKT_DEBUG_AMD") != nullptr;
  auto t_start = std::chrono::steady_clock::now();
  int dbg_next = 0; int64_t work_scan = 0;
  while (eliminated < nleft) {
    // the caller no longer needs this ordering (symbolic.cpp: the dissection candidate finished with a small top separator)
    if (cancel && cancel->load(std::memory_order_relaxed)) { order.clear(); return; }
    if (dbg && eliminated >= dbg_next) {
      fprintf(stderr, "okkt: amd %8d of %d eliminated, %zu pivots, %.3f s, mindeg %d, scanned %ld\n", eliminated, nleft, pivots.size(),
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), mindeg, (long)work_scan);
      dbg_next += nleft / 20;
    }
    // ---- pivot of least approximate degree
    while (mindeg < n && head[mindeg] < 0) ++mindeg;
    const int me = head[mindeg];
    dl_remove(me, mindeg);
    int nvpiv = nd[me].nv;
    // ---- room for the new element (its list has at most nd[me].degree entries)
    if (pfree + nd[me].degree + 1 > (int64_t)iw.size()) {
      compact();
      if (pfree + nd[me].degree + 1 + n > (int64_t)iw.size()) iw.resize((size_t)((pfree + nd[me].degree + n) * 3 / 2 + 64));
    }
    // ---- L_me = (A_me U union of L_e, e in E_me) \\ {me}: members are marked by nv < 0
    nd[me].nv = -nvpiv;
    const int64_t pme1 = pfree;
    int degme = 0;
    {
      const int64_t p0 = nd[me].pe;
      const int ne = nd[me].elen, nl = nd[me].len;
      for (int q = ne; q < nl; ++q) {            // variables adjacent to me
        const int v = iw[p0 + q];
        const int nvv = nd[v].nv;
        if (nvv > 0) { nd[v].nv = -nvv; iw[pfree++] = v; degme += nvv; dl_remove(v, nd[v].degree); }
      }
      for (int q = 0; q < ne; ++q) {             // elements adjacent to me are absorbed
        const int e = iw[p0 + q];
        if (nd[e].state != ST_ELEM) continue;
        const int64_t pe0 = nd[e].pe;
        for (int t = 0; t < nd[e].len; ++t) {
          const int v = iw[pe0 + t];
          const int nvv = nd[v].nv;
          if (nvv > 0) { nd[v].nv = -nvv; iw[pfree++] = v; degme += nvv; dl_remove(v, nd[v].degree); }
        }
        nd[e].state = ST_DEAD; nd[e].pe = -1; w[e] = 0;
      }
    }
    const int64_t pme2 = pfree;                  // L_me = iw[pme1, pme2)
    nd[me].state = ST_ELEM;
    nd[me].pe = pme1; nd[me].len = (int)(pme2 - pme1); nd[me].elen = 0;

    // ---- |L_e \\ L_me| for every element adjacent to a member of L_me
    for (int64_t q = pme1; q < pme2; ++q) {
      const int i = iw[q];
      const int nvi = -nd[i].nv;
      const int64_t p0 = nd[i].pe;
      work_scan += nd[i].len;
      for (int t = 0; t < nd[i].elen; ++t) {
        const int e = iw[p0 + t];
        const int64_t we = w[e];
        if (we >= wflg) w[e] = we - nvi;
        else if (we != 0) w[e] = (int64_t)nd[e].degree + wflg - nvi;     // nd[e].degree of an element = |L_e|
      }
    }
    // ---- degree update, list pruning, mass elimination, hashing
    touched.clear();
    for (int64_t q = pme1; q < pme2; ++q) {
      const int i = iw[q];
      const int64_t p0 = nd[i].pe;
      const int ne = nd[i].elen, nl = nd[i].len;
      int64_t deg = 0;
      uint64_t hash = 0;
      // surviving entries are gathered in a scratch list and written back as [me, elements ..., variables ...]; the list
      // loses at least one entry (me itself or an element that me absorbed), so it fits where it was
      keep.clear();
      int nelem = 0;
      for (int t = 0; t < ne; ++t) {
        const int e = iw[p0 + t];
        const int64_t we = w[e];
        if (we == 0) continue;                   // dead element
        const int64_t dext = we - wflg;
        if (dext > 0) { deg += dext; hash += (uint64_t)e; keep.push_back(e); ++nelem; }
        else { nd[e].state = ST_DEAD; nd[e].pe = -1; w[e] = 0; }           // aggressive absorption: L_e is inside L_me
      }
      for (int t = ne; t < nl; ++t) {
        const int v = iw[p0 + t];
        const int nvv = nd[v].nv;
        if (nvv > 0) { deg += nvv; hash += (uint64_t)v; keep.push_back(v); }
      }
      const int64_t pvars = p0 + 1 + nelem;
      const int64_t pn = p0 + 1 + (int64_t)keep.size();
      if (deg == 0 && nelem == 0 && pn == pvars) {
        // mass elimination: i has become indistinguishable from the pivot
        const int nvi = -nd[i].nv;
        nd[i].state = ST_ABSORBED; absorbed_into[i] = me;
        nvpiv += nvi; degme -= nvi;
        nd[i].nv = 0; nd[i].pe = -1; nd[i].len = 0; nd[i].elen = 0;
        continue;
      }
      // [me, e_1 .. e_k, e_0, variables]: the element that used to lead the list goes behind the other elements (the list order
      // decides ties between equal degrees further on; this is the order the round-1 implementation produced, kept so that
      // the orderings -- and the measured factorisations -- stay the same)
      iw[p0] = me;
      for (int t = 1; t < nelem; ++t) iw[p0 + (int64_t)t] = keep[t];
      if (nelem > 0) iw[p0 + nelem] = keep[0];
      for (size_t t = (size_t)nelem; t < keep.size(); ++t) iw[p0 + 1 + (int64_t)t] = keep[t];
      hash += (uint64_t)me;
      nd[i].elen = nelem + 1;
      nd[i].len = (int)(pn - p0);
      nd[i].degree = (int)std::min<int64_t>(nd[i].degree, deg);
      const int hk = (int)(hash % (uint64_t)n);
      if (hhead[hk] < 0) touched.push_back(hk);
      hnext[i] = hhead[hk];
      hhead[hk] = i;
    }
    // ---- supervariable detection among members of L_me with equal hash
    for (int hk : touched) {
      for (int i = hhead[hk]; i >= 0; i = hnext[i]) {
        if (nd[i].state != ST_VAR) continue;
        bool marked = false;
        const int64_t pi0 = nd[i].pe;
        for (int j = hnext[i]; j >= 0; j = hnext[j]) {
          if (nd[j].state != ST_VAR || nd[j].len != nd[i].len || nd[j].elen != nd[i].elen) continue;
          if (!marked) {                          // stamp i's entries once (slot 0 is `me` in every list of the bucket)
            ++stamp_tag;
            for (int t = 1; t < nd[i].len; ++t) stamp[iw[pi0 + t]] = stamp_tag;
            marked = true;
          }
          bool same = true;
          const int64_t pj0 = nd[j].pe;
          for (int t = 1; t < nd[j].len; ++t) if (stamp[iw[pj0 + t]] != stamp_tag) { same = false; break; }
          if (!same) continue;
          // j joins supervariable i (nv are negative here: both are marked members of L_me)
          nd[i].nv += nd[j].nv;
          nd[j].nv = 0;
          nd[j].state = ST_ABSORBED; absorbed_into[j] = i;
          nd[j].pe = -1; nd[j].len = 0; nd[j].elen = 0;
        }
      }
      hhead[hk] = -1;
    }
    // ---- finalise the element and re-insert the surviving members
    const int nleft_after = nleft - eliminated - nvpiv;
    int64_t pdst = pme1;
    for (int64_t q = pme1; q < pme2; ++q) {
      const int i = iw[q];
      if (nd[i].state != ST_VAR) continue;
      const int nvi = -nd[i].nv;
      nd[i].nv = nvi;
      iw[pdst++] = i;
      int64_t d = (int64_t)nd[i].degree + degme - nvi;
      d = std::min<int64_t>(d, (int64_t)nleft_after - nvi);
      if (d < 0) d = 0;
      nd[i].degree = (int)d;
      dl_insert(i, nd[i].degree);
    }
    nd[me].nv = 0;                                   // an element: never picked up as a variable again
    nd[me].len = (int)(pdst - pme1);
    nd[me].degree = degme;                           // |L_me| in columns, read as nd[e].degree above
    pfree = pdst;
    if (nd[me].len == 0) { nd[me].state = ST_DEAD; nd[me].pe = -1; w[me] = 0; }
    else w[me] = 1;                               // live element, not yet compared
    wflg += (int64_t)n + 1 + degme;               // every stale w[] entry stays below the new flag
    eliminated += nvpiv;
    pivots.push_back(me);
  }
  // ---- emit every pivot followed by everything ordered with it (the forest of absorbed variables below it)
  std::vector<int> cptr(n + 1, 0), clist;
  for (int i = 0; i < n; ++i) if (absorbed_into[i] >= 0) ++cptr[absorbed_into[i] + 1];
  for (int i = 0; i < n; ++i) cptr[i + 1] += cptr[i];
  clist.resize(cptr[n]);
  {
    std::vector<int> fill(cptr.begin(), cptr.end() - 1);
    for (int i = 0; i < n; ++i) if (absorbed_into[i] >= 0) clist[fill[absorbed_into[i]]++] = i;
  }
  for (int p : pivots) {
    const size_t start = order.size();
    order.push_back(p);
    for (size_t q = start; q < order.size(); ++q) {
      const int u = order[q];
      for (int t = cptr[u]; t < cptr[u + 1]; ++t) order.push_back(clist[t]);
    }
  }
  for (int d : dense_nodes) order.push_back(d);
}

}  // namespace okkt
