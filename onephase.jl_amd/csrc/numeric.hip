// Numeric supernodal multifrontal LDL^T (no pivoting) and the triangular solves, gfx950.
//
// Replaces, for the reference's hot path, CHOLMOD's numeric factorisation and solve that
// `cholesky(Symmetric(Q,:L))` / `ldlt(Symmetric(Q,:L))` / `F \ b` reach
// (/root/reference/src/linear_system_solvers/julia.jl:34,52,101,110).
//
// Fronts are dense f x f column-major buffers in one HBM arena (see symbolic.h).  The tree is
// processed level by level; inside a level
//   * fronts of order <= small_max are assembled, factored and written back by ONE workgroup
//     with the whole front resident in LDS (k_front_small),
//   * larger fronts are assembled by column blocks (k_big_assemble) and factored by a blocked
//     right-looking sweep: LDS diagonal block (k_big_diag), row-parallel triangular solve
//     (k_big_trsm) and an FP64-MFMA trailing update C -= W * L^T (k_big_syrk).
// Extend-add is deterministic: every destination entry is owned by exactly one workgroup that
// adds the children's contribution blocks in a fixed order -- no floating-point atomics.
#include "numeric.h"
#include "front_device.h"
#include <map>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace okkt {


#define OKKT_HIP_TRY(expr)                                                         \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess)                                                         \
      return std::string(#expr) + ": " + hipGetErrorString(e__);                   \
  } while (0)

// Retries only, once per level behind the small-front launches: the small fronts' slots are folded into slot 0, so the
// running totals that k_big_diag tests there are the true totals again, and the limits are tested on them.
__global__ void k_fold_counts(DevPlan P) {
  const int c = threadIdx.x;
  if (c >= 4) return;
  unsigned long long sum = 0;
  for (int q = 1; q < kCountSlots; ++q) {
    unsigned long long* w = P.counters + (size_t)q * kCountStride + c;
    sum += atomicExch(w, 0ull);     // two lanes may fold at the same time: every count is taken exactly once
  }
  const unsigned long long tot = atomicAdd(&P.counters[c], sum) + sum;
  const bool fail = c == 0 ? tot > (unsigned long long)P.want_pos : c == 1 ? tot > (unsigned long long)P.want_neg : tot > 0;
  if (fail) atomicExch(&P.counters[4], 1ull);
}

// Right-looking LDL^T of the leading npiv columns of an LDS-resident lower-triangular front.
// F is column-major with leading dimension ldf; wcol is scratch of length f.
template <int TPB>
__device__ __forceinline__ void ldl_partial_lds(double* F, double* wcol, int ldf, int f, int npiv) {
  constexpr int G = TPB / 32;
  const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
  for (int j = 0; j < npiv; ++j) {
    const double rd = 1.0 / F[j + j * ldf];  // one division per pivot; l = w * (1/d)
    for (int i = j + 1 + tid; i < f; i += TPB) {
      const double w = F[i + j * ldf];
      wcol[i] = w;             // w_i = l_ij * d_j
      F[i + j * ldf] = w * rd; // l_ij
    }
    __syncthreads();
    for (int c = j + 1 + grp; c < f; c += G) {
      const double wc = wcol[c];
      for (int i = c + lane; i < f; i += 32) F[i + c * ldf] -= F[i + j * ldf] * wc;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// small fronts: one workgroup, front resident in LDS
// ------------------------------------------------------------------------------------------
// FLOW (round 3): the tasks of SEVERAL levels in one launch.  `list` holds them level by level, so a task's children tasks have
// lower block indices (dispatch order = dependency order: a waiting workgroup never keeps its producers from being scheduled).
// A task raises flags[root] = epoch once its last front is in HBM; a parent waits on the flag of every child that lies outside
// its own index range.  The contribution blocks travel with agent-scope (sc1) accesses, which the memory side serves: no
// release / acquire fences.  A banded KKT system (BASELINE config 2) is factored by one launch instead of one per level.
__device__ __forceinline__ void flow_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double flow_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// bounded: a hand-off that never comes ends the wait instead of hanging the GPU -- and is counted as a non-finite pivot (`bad`,
// slot 0 word 3) and in the time-out word (slot 0 word 5): the factorisation reports a wrong inertia, solves return NaN
__device__ __forceinline__ void flow_wait(const int* flag, int value, unsigned long long* counters) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    if (spins >= (1 << 22)) { atomicAdd(&counters[3], 1ull); atomicExch(&counters[5], 1ull); }
  }
  __syncthreads();
}
__device__ __forceinline__ void flow_signal(int* flag, int value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int TPB, bool FLOW = false>
__global__ __launch_bounds__(TPB) void k_front_small(DevPlan P, const int* __restrict__ list, double tol, int* __restrict__ flags = nullptr, int epoch = 0, int wait_epoch = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int G = TPB / 32;
  const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
  const int s_root = list[blockIdx.x];
  unsigned pos = 0, neg = 0, zer = 0, bad = 0;
  if (stop_requested_wg(P)) {
    if (FLOW) flow_signal(flags + s_root, epoch);    // the consumers must not wait for a task that will never run
    return;
  }
  const int t_lo = P.task_lo[s_root];
  // the task: the fronts task_lo[root] .. root in postorder, children's contribution blocks pass through HBM (the
  // barrier at the end of an iteration makes the workgroup's stores visible to its own later loads)
  for (int s = t_lo; s <= s_root; ++s) {
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int ldf = f | 1;
  double* F = sm;
  double* wcol = sm + (size_t)ldf * f;
  double* front = P.arena + P.front_pos[s];

  for (int i = tid; i < ldf * f; i += TPB) F[i] = 0.0;
  __syncthreads();
  // original matrix entries of this supernode's columns
  {
    const int64_t e0 = P.aent_ptr[s], e1 = P.aent_ptr[s + 1];
    if (!P.has_dup) {
      for (int64_t e = e0 + tid; e < e1; e += TPB) {
        const int dst = P.aent_dst[e];
        const int lc = dst / f, lr = dst - lc * f;
        F[lr + lc * ldf] = P.vals[P.aent_src[e]];
      }
    } else if (tid == 0) {
      for (int64_t e = e0; e < e1; ++e) {
        const int dst = P.aent_dst[e];
        const int lc = dst / f, lr = dst - lc * f;
        F[lr + lc * ldf] += P.vals[P.aent_src[e]];
      }
    }
  }
  __syncthreads();
  for (int j = tid; j < k; j += TPB) F[j + j * ldf] += P.diagadd[col0 + j];
  __syncthreads();
  // extend-add of the children's contribution blocks, one child at a time (fixed order)
  for (int64_t q = P.child_ptr[s]; q < P.child_ptr[s + 1]; ++q) {
    const int c = P.children[q];
    const int kc = P.sn_col0[c + 1] - P.sn_col0[c];
    const int fc = (int)(P.row_ptr[c + 1] - P.row_ptr[c]);
    const int rc = fc - kc;
    const double* C = P.arena + P.front_pos[c] + P.cb_shift[c];      // only the child's contribution block (columns >= kc) is read
    const int* rl = P.rel + P.rel_ptr[c];
    if (FLOW && c < t_lo) flow_wait(flags + c, wait_epoch, P.counters);      // a child task of this launch (workgroup-uniform)
    for (int jj = grp; jj < rc; jj += G) {
      const int pj = rl[jj];
      const double* Ccol = C + (size_t)(kc + jj) * fc + kc;
      for (int ii = jj + lane; ii < rc; ii += 32) F[rl[ii] + pj * ldf] += FLOW ? flow_ld(Ccol + ii) : Ccol[ii];
    }
    __syncthreads();
  }
  ldl_partial_lds<TPB>(F, wcol, ldf, f, k);
  // write back: L panel (rows >= col), contribution block (lower), D, inertia counts
  for (int c = grp; c < f; c += G) {
    double* dst = front + (size_t)c * f + cb_off(P, s, c, k);
    if (FLOW && c >= k) { for (int i = c + lane; i < f; i += 32) flow_st(dst + i, F[i + c * ldf]); }
    else for (int i = c + lane; i < f; i += 32) dst[i] = F[i + c * ldf];
  }
  for (int j = tid; j < k; j += TPB) {
    const double d = F[j + j * ldf];
    P.dvals[col0 + j] = d;
    classify_pivot(d, tol, pos, neg, zer, bad);
  }
  if (FLOW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next front of the task reads this contribution block back
  __threadfence_block();
  __syncthreads();
  }
  flush_counts(P, 1 + (int)(blockIdx.x % (kCountSlots - 1)), pos, neg, zer, bad);
  if (FLOW) flow_signal(flags + s_root, epoch);
}

// ------------------------------------------------------------------------------------------
// big fronts: blocked right-looking LDL^T, block-column width NB (<= 128)
//   k_big_assemble : one wave per front column: zero, scatter A, extend-add via inverted lists
//   k_big_diag     : NB x NB diagonal block in LDS (inner width 32), D + inertia, and its inverse
//   k_big_trsm     : W = A21 * inv(L11)^T with FP64 MFMA, L21 = W * D^-1
//   k_big_syrk     : trailing update C -= W * L^T with FP64 MFMA, 128 x 128 tile per workgroup
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lower_bound_dev(const int* a, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

// Each front column is owned by exactly one wave, which applies the contributions in list order:
// deterministic, no atomics, no workgroup barriers.
// column pc of front s accumulated in `buf` (buf[0] = row pc): zero, A entries, delta, children's contributions in
// their fixed order.  LDS = true: buf is the wave's LDS column; false: the HBM column itself.
// extend-add items whose records and first rows are in flight together (round 5: packed 32-byte records, 4 / 8 / 16 items measured: 2047 / 2058 /
// 2485 us of assembly per S-metric factorisation -- the kernels are not bound by the record fetches)
#ifndef OKKT_ASM_BATCH
#define OKKT_ASM_BATCH 4
#endif
#ifndef OKKT_ASM_BATCH_CHUNKED
#define OKKT_ASM_BATCH_CHUNKED 4
#endif
#ifndef OKKT_ASM_UNROLL
#define OKKT_ASM_UNROLL 4
#endif
constexpr int kAsmBatch = OKKT_ASM_BATCH, kAsmBatchChunked = OKKT_ASM_BATCH_CHUNKED;
constexpr int kAsmUnroll = OKKT_ASM_UNROLL;      // 64-row groups of an item whose index and value loads are in flight together (round 6: 8 instead of 4 is SLOWER -- S-metric 17.3 against 17.1 ms, S-C5 3.95 / 3.87: the registers cost waves, and waves are what hides the latency here)
template <bool LDS, bool BATCH>
__device__ __forceinline__ void assemble_column(const DevPlan& P, int s, int pc, int f, int k, int col0, double* __restrict__ buf) {
  const int lane = threadIdx.x & 63;
  const int nrow = f - pc;
  for (int i = lane; i < nrow; i += 64) buf[i] = 0.0;
  __threadfence_block();
  if (pc < k) {
    const int64_t e0 = P.aent_ptr[s];
    const int ne = (int)(P.aent_ptr[s + 1] - e0);
    const int* dstv = P.aent_dst + e0;
    // range of this column's entries: precomputed on the host (two binary searches = ~25 dependent loads otherwise)
    const int64_t gcol = P.bigcol_base[s] + pc;
    const int lo = (int)(P.acol_lo[gcol] - e0);
    const int hi = pc + 1 < f ? (int)(P.acol_lo[gcol + 1] - e0) : ne;
    const int base = pc * f + pc;     // dstv holds offsets in the front; entry (row, pc) sits at pc * f + row
    if (!P.has_dup) {
      for (int e = lo + lane; e < hi; e += 64) buf[dstv[e] - base] = P.vals[P.aent_src[e0 + e]];
    } else if (lane == 0) {
      for (int e = lo; e < hi; ++e) buf[dstv[e] - base] += P.vals[P.aent_src[e0 + e]];
    }
    __threadfence_block();
    if (lane == 0) buf[0] += P.diagadd[col0 + pc];
    __threadfence_block();
  }
  const int64_t gc = P.bigcol_base[s] + pc;
  // one precomputed record per item (source column in the arena, rel list, lengths): the per-child metadata
  // would cost two more dependent memory round trips per item; the next record is fetched while the current
  // item is applied
  const int64_t q0 = P.ea_ptr[gc], q1 = P.ea_ptr[gc + 1];
  // Items are applied strictly in order (same destination column), but their loads need not wait for each other:
  // four items at a time have their records, and then the first 64 rows of their index and source columns, in
  // flight together (clamped addresses, no branches); most items of the lower levels are no longer than that.
  constexpr int NB4 = kAsmBatch;
  if constexpr (!BATCH) {
    // long columns (upper levels, bandwidth-bound): one item after the other, the next record fetched meanwhile
    int64_t src_n = 0, rel_n = 0;
    int rc_n = 0, jj_n = 0;
    if (q0 < q1) { const EaRec r = P.ea_rec[q0]; src_n = r.src; rel_n = r.rel; rc_n = r.rc; jj_n = r.jj; }
    for (int64_t q = q0; q < q1; ++q) {
      const int rc = rc_n, jj = jj_n;
      const int* rl = P.rel + rel_n;
      const double* Ccol = P.arena + src_n;
      if (q + 1 < q1) { const EaRec r = P.ea_rec[q + 1]; src_n = r.src; rel_n = r.rel; rc_n = r.rc; jj_n = r.jj; }
      int ii = jj + lane;
      for (; ii + 64 * (kAsmUnroll - 1) < rc; ii += 64 * kAsmUnroll) {
        int d[kAsmUnroll];
        double v[kAsmUnroll], o[kAsmUnroll];
#pragma unroll
        for (int u = 0; u < kAsmUnroll; ++u) d[u] = rl[ii + 64 * u] - pc;
#pragma unroll
        for (int u = 0; u < kAsmUnroll; ++u) { v[u] = Ccol[ii + 64 * u]; o[u] = buf[d[u]]; }
#pragma unroll
        for (int u = 0; u < kAsmUnroll; ++u) buf[d[u]] = o[u] + v[u];
      }
      for (; ii < rc; ii += 64) buf[rl[ii] - pc] += Ccol[ii];
      __threadfence_block();
    }
    return;
  }
  for (int64_t q = q0; q < q1; q += NB4) {
    int64_t src[NB4], rel[NB4];
    int rc[NB4], jj[NB4];
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      const EaRec r = P.ea_rec[min(q + u, q1 - 1)];
      src[u] = r.src; rel[u] = r.rel; rc[u] = r.rc; jj[u] = r.jj;
    }
    int d[NB4];
    double v[NB4];
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      const int ii = min(jj[u] + lane, rc[u] - 1);
      d[u] = P.rel[rel[u] + ii] - pc;
      v[u] = P.arena[src[u] + ii];
    }
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      if (q + u < q1) {       // wave-uniform
        if (jj[u] + lane < rc[u]) buf[d[u]] += v[u];
        const int* rl = P.rel + rel[u];
        const double* Ccol = P.arena + src[u];
        int ii = jj[u] + 64 + lane;
        for (; ii + 64 * (kAsmUnroll - 1) < rc[u]; ii += 64 * kAsmUnroll) {
          int dd[kAsmUnroll];
          double vv[kAsmUnroll], oo[kAsmUnroll];
#pragma unroll
          for (int w = 0; w < kAsmUnroll; ++w) dd[w] = rl[ii + 64 * w] - pc;
#pragma unroll
          for (int w = 0; w < kAsmUnroll; ++w) { vv[w] = Ccol[ii + 64 * w]; oo[w] = buf[dd[w]]; }
#pragma unroll
          for (int w = 0; w < kAsmUnroll; ++w) buf[dd[w]] = oo[w] + vv[w];
        }
        for (; ii < rc[u]; ii += 64) buf[rl[ii] - pc] += Ccol[ii];
        __threadfence_block();
      }
    }
  }
}

template <bool BATCH>
__global__ __launch_bounds__(256) void k_big_assemble(DevPlan P, const int* __restrict__ list, int lcol) {
  // One wave per front column.  Columns of at most `lcol` rows are accumulated in LDS and written to HBM once:
  // the read-modify-write chain of the extend-add then runs at LDS latency instead of an HBM round trip per
  // item, and the zero-fill and the read-back of the column never reach HBM.  Longer columns take the same steps
  // directly in HBM.  The summation order is the same on both paths (bitwise identical results).
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = list[blockIdx.y];
  if (stop_requested_wave(P)) return;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int pc = blockIdx.x * 4 + wv;
  if (pc >= f) return;
  double* col = P.arena + P.front_pos[s] + (size_t)pc * f + pc + cb_off(P, s, pc, k);   // first row of the lower triangle: (pc, pc)
  const int nrow = f - pc;
  if (nrow <= lcol) {
    double* buf = sm + (size_t)wv * lcol;
    assemble_column<true, BATCH>(P, s, pc, f, k, col0, buf);
    for (int i = lane; i < nrow; i += 64) col[i] = buf[i];
  } else {
    assemble_column<false, BATCH>(P, s, pc, f, k, col0, col);
  }
}

// Upper levels (fronts with more than 2048 rows): the column is cut into row chunks of kAsmChunk rows and each
// (column, chunk) is a wave with its chunk in LDS -- every front entry is written to HBM exactly once and every
// child entry read once (the in-HBM path zero-fills, then reads and writes the column once per contributing item).
// Per child a small table gives the position in its rel list where each chunk boundary falls (the list is sorted),
// and per front column the range of its A entries is precomputed: no searches on the device.
#ifndef OKKT_ASM_CHUNK
#define OKKT_ASM_CHUNK 1024
#endif
constexpr int kAsmChunk = OKKT_ASM_CHUNK;
__global__ __launch_bounds__(256) void k_big_assemble_chunked(DevPlan P, const int* __restrict__ list) {
  __shared__ double sm[4 * kAsmChunk];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = list[blockIdx.y];
  if (stop_requested_wave(P)) return;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int pc = blockIdx.x * 4 + wv;
  if (pc >= f) return;
  const int t = blockIdx.z;
  const int r0 = max(pc, t * kAsmChunk), r1 = min(f, (t + 1) * kAsmChunk);
  if (r0 >= r1) return;
  double* buf = sm + wv * kAsmChunk - r0;      // buf[row] for row in [r0, r1): LDS array (static, so the bias is harmless)
  double* base = sm + wv * kAsmChunk;
  const int n = r1 - r0;
  for (int i = lane; i < n; i += 64) base[i] = 0.0;
  __threadfence_block();
  const int64_t gc = P.bigcol_base[s] + pc;
  if (pc < k) {
    const int64_t lo = P.acol_lo[gc], hi = pc + 1 < f ? P.acol_lo[gc + 1] : P.aent_ptr[s + 1];
    const int cbase = pc * f;
    if (!P.has_dup) {
      for (int64_t e = lo + lane; e < hi; e += 64) {
        const int row = P.aent_dst[e] - cbase;
        if (row >= r0 && row < r1) base[row - r0] = P.vals[P.aent_src[e]];
      }
    } else if (lane == 0) {
      for (int64_t e = lo; e < hi; ++e) {
        const int row = P.aent_dst[e] - cbase;
        if (row >= r0 && row < r1) base[row - r0] += P.vals[P.aent_src[e]];
      }
    }
    __threadfence_block();
    if (lane == 0 && pc >= r0 && pc < r1) base[pc - r0] += P.diagadd[col0 + pc];
    __threadfence_block();
  }
  (void)buf;
  const int64_t q0 = P.ea_ptr[gc], q1 = P.ea_ptr[gc + 1];
  // Items are applied strictly in list order (same summation order as every other assembly path: bitwise identical results),
  // but four at a time have their records, then their chunk boundaries, then the first 64 rows of their index and value
  // columns in flight together (clamped addresses, no branches around the loads): three dependent round trips per four
  // items instead of two per item -- the kernel is pure memory latency (SQ_WAIT_ANY 85 % of the wave cycles).
  constexpr int NB4 = kAsmBatchChunked;
  for (int64_t q = q0; q < q1; q += NB4) {
    int64_t src[NB4], rel[NB4], cutp[NB4];
    int rc[NB4], jj[NB4], lo[NB4], hi[NB4];
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      const EaRec r = P.ea_rec[min(q + u, q1 - 1)];
      src[u] = r.src; rel[u] = r.rel; rc[u] = r.rc; jj[u] = r.jj; cutp[u] = r.cut;
    }
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      const int* cut = P.cutv + cutp[u];
      lo[u] = max(jj[u], cut[t]); hi[u] = min(rc[u], cut[t + 1]);      // rows of this item that fall into the chunk
    }
    int d[NB4];
    double v[NB4];
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      const int ii = min(lo[u] + lane, rc[u] - 1);
      d[u] = P.rel[rel[u] + ii] - r0;
      v[u] = P.arena[src[u] + ii];
    }
#pragma unroll
    for (int u = 0; u < NB4; ++u) {
      if (q + u < q1) {       // wave-uniform
        if (lo[u] + lane < hi[u]) base[d[u]] += v[u];
        const int* rl = P.rel + rel[u];
        const double* Ccol = P.arena + src[u];
        int ii = lo[u] + 64 + lane;
        for (; ii + 64 * (kAsmUnroll - 1) < hi[u]; ii += 64 * kAsmUnroll) {
          int dd[kAsmUnroll];
          double vv[kAsmUnroll], oo[kAsmUnroll];
#pragma unroll
          for (int w = 0; w < kAsmUnroll; ++w) dd[w] = rl[ii + 64 * w] - r0;
#pragma unroll
          for (int w = 0; w < kAsmUnroll; ++w) { vv[w] = Ccol[ii + 64 * w]; oo[w] = base[dd[w]]; }
#pragma unroll
          for (int w = 0; w < kAsmUnroll; ++w) base[dd[w]] = oo[w] + vv[w];
        }
        for (; ii < hi[u]; ii += 64) base[rl[ii] - r0] += Ccol[ii];
        __threadfence_block();
      }
    }
  }
  double* col = P.arena + P.front_pos[s] + (size_t)pc * f + r0 + cb_off(P, s, pc, k);
  for (int i = lane; i < n; i += 64) col[i] = base[i];
}




// NB x NB diagonal block of block-column `step`: LDL^T in LDS (inner width 32: register/readlane
// 32 x 32 kernel on one wave, row-parallel solve below it, MFMA rank-32 update), D and inertia out,
// then X = inv(L11) block by block (MFMA products) for the row-parallel k_big_trsm and the solves.

// k_big_diag: LDL^T of the NB x NB diagonal block held in MFMA ACCUMULATORS.
//   The block is viewed as 8 x 8 tiles of 16 x 16 (identity-padded past nb); the 36 lower tiles are dealt
//   cyclically to the four waves (9 accumulator quads per lane).  A micro-step factors 8 columns:
//     (1) the waves that own the tile column copy the 8 columns out of their accumulators into LDS (Praw),
//     (2) threads 0..127 (one per row) each redo the 8 x 8 diagonal LDL^T in registers (broadcast LDS reads, no
//         communication: the 8 dependent reciprocals are the irreducible latency) and solve their own row,
//         write the final L entries to HBM (and the 32 x 32 diagonal blocks to LDS for the inverse) and the
//         operand panels -L, W = L D to LDS,
//     (3) every live tile gets its rank-8 update as two v_mfma_f64_16x16x4 straight into the accumulators.
//   16 micro-steps x 2 barriers replace the 128 pivot round trips of a row-on-lane LDS version (80 us -> 42 us per block).
//   Tile entry held by lane (l15, l4), register v of tile (ti, tj):  row 16 ti + l15, column 16 tj + 4 v + l4
//   (rows on l15: the initial load reads 128-byte row segments of the column-major front).
//   Entries above the diagonal and columns already factored are dead: the updates may write garbage there.
//   Finally wave b inverts the unit-lower 32 x 32 diagonal block b: one COLUMN of X per lane, forward
//   substitution with broadcast LDS reads of L (no cross-lane traffic at all).

__global__ __launch_bounds__(256) void k_big_diag(DevPlan P, const int* __restrict__ list, int step, int NB, double tol, int dbg_stop) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int s = list[blockIdx.x];
  if (stop_requested_wg(P)) return;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = step * NB;
  if (j0 >= k) return;
  const int nb = min(NB, k - j0);
  double* Praw = sm;                         // 8 x kPLD: raw micro-panel (columns as rows of the array)
  double* Lp = Praw + kMW * kPLD;            // -L of the micro-panel
  double* Wp = Lp + kMW * kPLD;              // W = L * D
  double* Ld = Wp + kMW * kPLD;              // 4 diagonal 32 x 32 blocks of L, leading dimension 33
  double* Xs = Ld + 4 * 32 * kXld;           // their inverses
  double* F = P.arena + P.front_pos[s];

  // tiles of this wave: t = 4 s + wave in column-major order of the lower triangle of the 8 x 8 tile grid
  int ti_s[9], tj_s[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int t = 4 * q + wave;
    const int tj = (t >= 8) + (t >= 15) + (t >= 21) + (t >= 26) + (t >= 30) + (t >= 33) + (t >= 35);
    const int start = tj * 8 - tj * (tj - 1) / 2;
    tj_s[q] = tj;
    ti_s[q] = tj + (t - start);
  }
  long long tdbg0 = dbg_stop == 9 ? wall_clock64() : 0, tA = 0, tB = 0, tC = 0, tL = 0;
  d4_t acc[9];
  {
    // all loads first (clamped addresses, no branches), then the selects
    double raw[9][4];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int r = 16 * ti_s[q] + l15;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c = 16 * tj_s[q] + 4 * v + l4;
        raw[q][v] = F[(size_t)(j0 + min(c, nb - 1)) * f + j0 + min(r, nb - 1)];
      }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int r = 16 * ti_s[q] + l15;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c = 16 * tj_s[q] + 4 * v + l4;
        const double pad = r == c ? 1.0 : 0.0;
        acc[q][v] = (r < nb && c < nb && r >= c) ? raw[q][v] : pad;
      }
    }
  }
  if (dbg_stop == 1) return;
  if (dbg_stop == 9) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tL = wall_clock64() - tdbg0; }
  double my_d = 1.0;
  const int npair = (nb + 15) >> 4;
  for (int pp = 0; pp < npair; ++pp) {
#pragma unroll
    for (int h = 0; h < 16 / kMW; ++h) {
      const int p8 = pp * 16 + h * kMW;      // first column of the micro-panel
      if (p8 >= nb) continue;                // uniform
      long long tq = dbg_stop == 9 ? wall_clock64() : 0;
      // (1) copy columns [p8, p8 + 8) out of the accumulators
#pragma unroll
      for (int q = 0; q < 9; ++q)
        if (tj_s[q] == pp) {
          const int r = 16 * ti_s[q] + l15;
#pragma unroll
          for (int e = 0; e < kMW / 4; ++e) Praw[(4 * e + l4) * kPLD + r] = acc[q][h * (kMW / 4) + e];
        }
      // LDS-only barrier: __syncthreads() would also wait for the global stores of L (an HBM round trip per micro-step)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (dbg_stop == 9) { const long long tn = wall_clock64(); tA += tn - tq; tq = tn; }
      // (2) one thread per row
      if (tid < 128 && tid >= p8 && dbg_stop != 3) {
        const int r = tid;
        double A[kMW][kMW], a[kMW], rd[kMW], w[kMW], lr[kMW];
#pragma unroll
        for (int c = 0; c < kMW; ++c) {
          a[c] = Praw[c * kPLD + r];
#pragma unroll
          for (int i = c; i < kMW; ++i) A[i][c] = Praw[c * kPLD + p8 + i];
        }
#pragma unroll
        for (int c = 0; c < kMW; ++c) {
          rd[c] = fast_rcp_f64(A[c][c]);
          w[c] = a[c];
          lr[c] = w[c] * rd[c];
#pragma unroll
          for (int i = c + 1; i < kMW; ++i) {
            const double lic = A[i][c] * rd[c];
#pragma unroll
            for (int j = c + 1; j <= i; ++j) A[i][j] = __builtin_fma(-lic, A[j][c], A[i][j]);
          }
#pragma unroll
          for (int j = c + 1; j < kMW; ++j) a[j] = __builtin_fma(-lr[c], A[j][c], a[j]);
        }
        const int i = r - p8;                // >= 0; < 8: a row of the diagonal block
        double* Fr = F + (size_t)(j0 + p8) * f + j0 + r;
#pragma unroll
        for (int c = 0; c < kMW; ++c) {
          Lp[c * kPLD + r] = -lr[c];
          Wp[c * kPLD + r] = w[c];
          // final entry (r, p8 + c): L below the diagonal, D on it, nothing above
          const double val = i == c ? w[c] : lr[c];
          if (i >= c && r < nb && p8 + c < nb) {
            Fr[(size_t)c * f] = val;
            if ((r >> 5) == ((p8 + c) >> 5)) Ld[(r >> 5) * 32 * kXld + (r & 31) + ((p8 + c) & 31) * kXld] = val;
          }
        }
        if (i < kMW) {   // each row is a pivot row in exactly one micro-step: classified once, after the loop
#pragma unroll
          for (int c = 0; c < kMW; ++c) my_d = i == c ? w[c] : my_d;
        }
      }
      // LDS-only barrier: __syncthreads() would also wait for the global stores of L (an HBM round trip per micro-step)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (dbg_stop == 9) { const long long tn = wall_clock64(); tB += tn - tq; tq = tn; }
      // (3) rank-8 update of every tile that still has live columns (to the right of the micro-panel)
#pragma unroll
      for (int q = 0; q < 9; ++q)
        if (tj_s[q] * 16 + 15 >= p8 + kMW && dbg_stop != 4) {
          const int rr = 16 * ti_s[q] + l15, cc = 16 * tj_s[q] + l15;
          double av[kMW / 4], bv[kMW / 4];
#pragma unroll
          for (int e = 0; e < kMW / 4; ++e) { av[e] = Wp[(4 * e + l4) * kPLD + cc]; bv[e] = Lp[(4 * e + l4) * kPLD + rr]; }
#pragma unroll
          for (int e = 0; e < kMW / 4; ++e) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[e], bv[e], acc[q], 0, 0, 0);
        }
      if (dbg_stop == 9) { tC += wall_clock64() - tq; }
    }
  }
  {
    unsigned pos = 0, neg = 0, zer = 0, bad = 0;
    if (tid < nb) {
      P.dvals[col0 + j0 + tid] = my_d;
      classify_pivot(my_d, tol, pos, neg, zer, bad);
    }
    if (dbg_stop != 9) flush_counts(P, 0, pos, neg, zer, bad);
    else if (tid == 0) {   // debug: phase times (wall-clock ticks x 1000) instead of pivot counts
      atomicAdd(&P.counters[0], (unsigned long long)tL * 1000ull); atomicAdd(&P.counters[1], (unsigned long long)tA * 1000ull);
      atomicAdd(&P.counters[2], (unsigned long long)tB * 1000ull); atomicAdd(&P.counters[3], (unsigned long long)tC * 1000ull);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (dbg_stop == 2) return;
  // X_bb = inv(L_bb): wave b, one column per lane
  const int off = wave * 32;
  if (off < nb) {
    const int w = min(32, nb - off);
    const double* Lb = Ld + wave * 32 * kXld;
    double* Xb = Xs + wave * 32 * kXld;
    if (lane < 32) {
      const int c = lane;
      double x[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        double v = (r == c) ? 1.0 : 0.0;
        if (r < w) {
#pragma unroll
          for (int p = 0; p < r; ++p) v = __builtin_fma(-Lb[r + p * kXld], x[p], v);
        }
        x[r] = (r < w && c < w && r >= c) ? v : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 32; ++r) Xb[r + c * kXld] = x[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    double* Xg = P.invl + P.invl_pos[s] + (size_t)step * NB * NB;
    const int r = lane & 31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = (lane >> 5) + 2 * q;
      Xg[(off + r) + (size_t)(off + c) * NB] = Xb[r + c * kXld];
    }
  }
}

__global__ __launch_bounds__(384) void k_big_diag2(DevPlan P, const int* __restrict__ list, int step, int NB, double tol) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int s = list[blockIdx.x];
  if (stop_requested_wg(P)) return;
  diag2_body<false>(P, s, step, NB, tol, sm);
}


// Deferred full inverse X = inv(L11) of every NB x NB diagonal block of the big fronts of a level (one
// workgroup per block, all blocks in parallel, off the critical path of the factorisation: k_big_trsm
// only needs the 32 x 32 diagonal inverses X_ii that k_big_diag leaves in `invl`).  The blocked solves read X.
// Off-diagonal blocks X_ij = -X_ii * (sum_{p=j}^{i-1} L_ip X_pj) by block distance, 32 x 32 MFMA products.
__global__ __launch_bounds__(256) void k_big_invert(DevPlan P, const int* __restrict__ list, int NB, int step0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int s = list[blockIdx.y];
  const int step = step0 + (int)blockIdx.x;
  if (stop_requested_wg(P)) return;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = step * NB;
  if (j0 >= k) return;
  const int nb = min(NB, k - j0);
  const int ldb = NB + 2;
  double* B = sm;
  double* T = sm + (size_t)ldb * NB;   // 3 scratch blocks, leading dimension kTld
  const double* F = P.arena + P.front_pos[s];
  const int nblk = (nb + kIB - 1) / kIB;
  double* X = P.invl + P.invl_pos[s] + (size_t)step * NB * NB;
  {
    const int b = tid >> 6, cc = (tid >> 1) & 31, h = tid & 1;
    if (b < nblk) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = X[(b * kIB + h * 16 + q) + (size_t)(b * kIB + cc) * NB];
#pragma unroll
      for (int q = 0; q < 16; ++q) B[(b * kIB + h * 16 + q) + (size_t)(b * kIB + cc) * ldb] = v[q];
    }
  }
  __syncthreads();
  for (int d = 1; d < nblk; ++d) {
    const int npairs = nblk - d;
    for (int t = wave; t < npairs * 4; t += 4) {
      const int bj = t >> 2, bi = bj + d, sub = t & 3;
      const int ro = bi * kIB, co = bj * kIB;
      const int rr0 = (sub & 1) * 16, cc0 = (sub >> 1) * 16;
      d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
      const int lrow = ro + rr0 + l15;
      double avv[24];
      const int lrc = min(lrow, nb - 1);
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        const int p = min(co + 4 * q + l4, nb - 1);
        avv[q] = keep_f64(F[(size_t)(j0 + p) * f + j0 + lrc], lrow < nb && co + 4 * q < ro);
      }
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        const int p0 = co + 4 * q;
        if (p0 < ro) {
          const double bv = B[(p0 + l4) + (size_t)(co + cc0 + l15) * ldb];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(avv[q], bv, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        T[bj * kTld * kIB + (rr0 + l4 + 4 * reg) + (cc0 + l15) * kTld] = acc[reg];
    }
    __syncthreads();
    for (int t = wave; t < npairs * 4; t += 4) {
      const int bj = t >> 2, bi = bj + d, sub = t & 3;
      const int ro = bi * kIB, co = bj * kIB;
      const int rr0 = (sub & 1) * 16, cc0 = (sub >> 1) * 16;
      d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int p0 = 0; p0 < kIB; p0 += 4) {
        const double av = B[(ro + rr0 + l15) + (size_t)(ro + p0 + l4) * ldb];
        const double bv = T[bj * kTld * kIB + (p0 + l4) + (cc0 + l15) * kTld];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        B[(ro + rr0 + l4 + 4 * reg) + (size_t)(co + cc0 + l15) * ldb] = -acc[reg];
    }
    __syncthreads();
  }
  for (int c = wave; c < NB; c += 4) {
    const int bc = c / kIB;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = 2 * lane + u;
      if (i < NB && i / kIB != bc) X[i + (size_t)c * NB] = (i < nb && c < nb && i / kIB > bc) ? B[i + (size_t)c * ldb] : 0.0;
    }
  }
}

template <int NBLK>
__global__ __launch_bounds__(256) void k_big_trsm(DevPlan P, const int* __restrict__ list, int step, int wcol0, int blk_lo) {
  // blk_lo: first 64-row block of the panel this launch handles (the grid covers blocks blk_lo, blk_lo + 1, ...)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int s = list[blockIdx.y];
  if (stop_requested_wg(P)) return;
  trsm_body<NBLK, false>(P, s, step, wcol0, (int)blockIdx.x + blk_lo, sm, nullptr, 0);
}
// k_big_diag2 and k_big_trsm<4> of one panel step in ONE launch (round 3): workgroup 0 of a front factors the diagonal block
// and raises the front's flag; the others have their panel rows in flight by then, wait, stage the block and solve.  What
// crosses workgroups inside the launch (L11, its 32 x 32 inverses, D) is written and read with agent-scope accesses.
__global__ __launch_bounds__(384) void k_diag_trsm_fused(DevPlan P, const int* __restrict__ list, int step, int NB, double tol, int wcol0,
                                                         int* __restrict__ flags, int epoch, int cnt, int ntr, int drop) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  // one-dimensional grid, the diagonal blocks of ALL fronts first (cnt workgroups), then the trsm workgroups front by front: with
  // (1 + ntr, cnt) the waiting trsm workgroups of the first fronts filled the CUs before the other fronts' diagonal blocks were
  // dispatched (one 110-KB workgroup per CU) and a level of 50 fronts took three rounds
  const int x = (int)blockIdx.x;
  const int s = list[x < cnt ? x : (x - cnt) / ntr];
  if (stop_requested_wg(P)) {
    // the trsm workgroups of this front may have passed their own test before the flag was raised: they must not wait for a block
    // that will never be factored (advisor, round 3)
    if (x < cnt && threadIdx.x == 0) __hip_atomic_store(flags + s, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (x < cnt) {
    diag2_body<true>(P, s, step, NB, tol, sm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its agent-scope stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + s, epoch - drop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    trsm_body<4, true>(P, s, step, wcol0, (x - cnt) % ntr, sm, flags + s, epoch);
  }
}

// trailing update  F[r, c] -= sum_p W[r, p] * L[c, p]  for r >= c >= j0 + nb  (the dominant kernel).
//
// 128 x 128 tile per workgroup (one workgroup per CU), 64 x 64 per wave.
//  * MFMA shape: v_mfma_f64_4x4x4_4b_f64.  Measured on MI355X (scripts/mfma_f64_peak2.hip) it sustains
//    68-76 TFLOP/s, the 16x16x4 form only 36-50.  Lane l of the instruction holds A[i = l&3][k = l>>4] and
//    B[k = l>>4][j = l&3] of block (l>>2)&3 and receives D[i = l>>4][j = l&3] of that block (decoded with
//    scripts/mfma_f64_4x4_decode.hip).  Feeding the same 4 panel columns c to all four A blocks (an LDS
//    broadcast read) and 16 consecutive front rows r to B turns one instruction into a 16-row x 4-column
//    strip of C: D lane l <-> row l&15, column l>>4, so C moves in whole 128-byte segments.
//    A is negated by the instruction's neg modifier, so the accumulators simply start from C.
//  * operands: both panels are staged by LDS-DMA (global_load_lds_dwordx4, one 1-KiB panel row per wave
//    instruction, no staging registers) into a ring of three 16-column chunks; two chunks stay in flight
//    across the (raw) barrier behind a counted s_waitcnt vmcnt.  LDS rows are padded to 144 doubles so the
//    four k-slices of a B fragment fall on disjoint banks.
//  * tile order: workgroups b, b+8, .. share an XCD; they get consecutive tiles (same W row block in L2).

// TC = tile width in columns (round 3): 128, or 64 for trailing updates of few tiles -- a launch that cannot give every CU two
// workgroups of 128 x 128 (<= 496 tiles: the chain-bound tail of every front, 45 us per 128-column step whatever is left)
// runs as 128 x 64 tiles: twice the workgroups with half the work each, all four SIMDs of a CU busy.  The L operand is still
// staged 128 rows deep (the DMA shape stays), only the first 64 are read.
template <int DBG, int HEAD, int TC = 128>
__global__ __launch_bounds__(kSyrkNW * 64, kSyrkNW / 2) void k_big_syrk(DevPlan P, const int* __restrict__ list, int stepA, int npan,
                                                                        int tstep, int NB, int wofs, int csplit) {
  static_assert(TC == 128 || (TC == 64 && HEAD == kSyrkTrail), "64-column tiles exist for the trailing triangle only");
  // Applies the panels [stepA, stepA + npan) (K = up to npan * NB columns, W panels side by side in wbuf from
  // column wofs) to the region that starts at block column tstep; K = GS * NB halves the C traffic per flop.
  constexpr int NW = kSyrkNW, STAGES = kSyrkStages;
  constexpr int DMA = 2 * (kSyrkKC / NW);   // LDS-DMA instructions per wave and chunk
  constexpr int WCW = TC / (NW / 2);    // columns per wave
  constexpr int NCG = WCW / 4;          // 4-column groups per wave
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tkA = 0, tkB = 0;
  if constexpr (DBG & 16) tk0 = wall_clock64();
  const int s = list[blockIdx.y];
  if (stop_requested_wg(P)) return;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = stepA * NB;
  if (j0 >= k) return;
  const int nb = min(npan * NB, k - j0);          // K of this update
  if (HEAD == kSyrkPanel && tstep * NB >= k) return;   // there is no next panel in this front
  const int t0 = min(tstep * NB, k);
  const int climit = HEAD == kSyrkPanel ? min(t0 + NB, k) : f;  // columns this launch may write
  const int T = (f - t0 + 127) >> 7;
  int ntiles;
  if (HEAD == kSyrkTrail) { const int Tr = max(T - csplit, 0); ntiles = TC == 128 ? Tr * (Tr + 1) / 2 : Tr * (Tr + 1); }   // 64-wide: row tile ti has 2 ti + 2 tiles
  else if (HEAD == kSyrkPanel) ntiles = csplit == 1 ? min(T, 1) : (csplit == 2 ? max(T - 1, 0) : T);   // all / diagonal tile / the rest
  else { ntiles = 0; for (int c = 0; c < csplit && c < T; ++c) ntiles += T - c; }
  // XCD-aware order: consecutive tile indices (which share operand panels) stay on one XCD and its L2
  // `per` from THIS front's tile count, not from the grid (which is sized for the largest front of the level):
  // otherwise a smaller front of a batched launch runs on the first one or two XCDs only
  const int per = (ntiles + 7) >> 3;
  if ((int)(blockIdx.x >> 3) >= per) return;
  int idx = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (idx >= ntiles) return;
  int ti, tj;
  if (HEAD == kSyrkPanel) { ti = idx + (csplit == 2 ? 1 : 0); tj = 0; }
  else if (HEAD == kSyrkAhead) {
    int rest = idx; tj = 0;
    while (rest >= T - tj) { rest -= T - tj; ++tj; }
    ti = tj + rest;
  } else if (TC == 128) {
    ti = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= idx) ++ti;
    while (ti * (ti + 1) / 2 > idx) --ti;
    tj = idx - ti * (ti + 1) / 2;
    ti += csplit; tj += csplit;
  } else {
    ti = (int)((sqrtf(4.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) <= idx) ++ti;
    while (ti * (ti + 1) > idx) --ti;
    tj = idx - ti * (ti + 1);             // 0 .. 2 ti + 1, in 64-column units
    ti += csplit; tj += 2 * csplit;
  }
  const int rt0 = t0 + ti * 128, ct0 = t0 + tj * TC;   // tile origin
  const int rbase = rt0 + (wv & 1) * 64;
  const int cbase = ct0 + (wv >> 1) * WCW;
  const bool active = !(rbase + 63 < cbase) && rbase < f && cbase < climit;
  double* F = P.arena + P.front_pos[s];
  const int kfront = P.sn_col0[s + 1] - P.sn_col0[s];      // columns from here on live in the shared contribution-block region (cb_off)
  const double* Wg = P.wbuf + P.wbuf_pos[s] + (size_t)wofs * f + rt0 + lane * 2;
  const double* Lg = F + (size_t)j0 * f + ct0 + lane * 2;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nchunk = (nb + kSyrkKC - 1) / kSyrkKC;

  // stage `ch` -> ring slot ch % 2: wave wv moves panel rows q * 8 + wv (q = 0, 1) of both operands by LDS-DMA
  auto issue = [&](int ch) {
    double* slot = sm + (size_t)(ch % STAGES) * 2 * kSyrkKC * kSyrkLd;
#pragma unroll
    for (int q = 0; q < kSyrkKC / NW; ++q) {
      const int prow = q * NW + wv;
      const int p = ch * kSyrkKC + prow;
      // rows past the panel read a zero page instead (same instruction count on every path)
      const double* wsrc = p < nb ? Wg + (size_t)p * f : P.zero_page + lane * 2;
      const double* lsrc = p < nb ? Lg + (size_t)p * f : P.zero_page + lane * 2;
      __builtin_amdgcn_global_load_lds(wsrc, (lds_void_t*)(slot + prow * kSyrkLd), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(lsrc, (lds_void_t*)(slot + (kSyrkKC + prow) * kSyrkLd), 16, 0, 0);
    }
  };

  if constexpr (DBG & 16) { tkA = wall_clock64(); }
  double acc[NCG][4];  // [column group of 4][row block of 16]
  // accumulators start from C (guarded; lanes outside the front or above the diagonal hold zeros)
  // Loads are unconditional on clamped (always valid) addresses and selected afterwards: a load under
  // a per-element branch makes hipcc wait vmcnt(0) per element, i.e. 64 serial memory round trips.
  // Row map of an accumulator: (l15, rb) <-> row 2*l15 + (rb & 1) + 32*(rb >> 1) of the wave's 64 rows, so
  // that a lane's (rb, rb+1) pair is 16 contiguous bytes of C: 16-byte loads/stores, 256-byte segments.
  typedef double d2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    const int c = cbase + cg * 4 + l4;
    const double* colp = F + (size_t)min(c, f - 1) * f + cb_off(P, s, min(c, f - 1), kfront);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = rbase + 2 * l15 + 32 * h;
      const int rcl = min(r, f - 2);            // clamped pair start: always in bounds, no branch
      const int shift = r - rcl;                // 0 in the interior, 1 when r is the last row, >= 2 outside
      d2_t v = (d2_t){0.0, 0.0};
      if constexpr (!(DBG & 1)) __builtin_memcpy(&v, colp + rcl, 16);
      const double e0 = shift == 0 ? v[0] : v[1];
      acc[cg][2 * h] = keep_f64(e0, shift <= 1 && c < climit && r >= c);
      acc[cg][2 * h + 1] = keep_f64(v[1], shift == 0 && c < climit && r + 1 >= c);
    }
  }
  // the LDS-DMAs go out AFTER the C loads: with an LDS-DMA in flight hipcc waits vmcnt(0) after every
  // ordinary load (64 serial round trips); in this order it issues all 64 back to back
  asm volatile("" ::: "memory");
  if constexpr (DBG & 32) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tkB = wall_clock64(); }
  if constexpr (!(DBG & 8)) {
#pragma unroll
    for (int q = 0; q < STAGES - 1; ++q) if (q < nchunk) issue(q);
  }
  for (int ch = 0; ch < nchunk; ++ch) {
    // chunk ch has landed once only the chunks issued after it (at most STAGES - 2) are still outstanding
    if constexpr (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
      const int later = min(STAGES - 2, nchunk - 1 - ch);
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (DBG & 16) { if (ch == 0) tk1 = wall_clock64(); }   // C tile and the first operand chunk have landed
    // the slot written next was last read one iteration ago; everyone is past that barrier
    if constexpr (!(DBG & 8)) { if (ch + STAGES - 1 < nchunk) issue(ch + STAGES - 1); }
    if (active && !(DBG & 2)) {
      const double* slot = sm + (size_t)(ch % STAGES) * 2 * kSyrkKC * kSyrkLd;
      const double* bw = slot + (wv & 1) * 64 + 2 * l15;
      const double* bl = slot + kSyrkKC * kSyrkLd + (wv >> 1) * WCW + (lane & 3);
      // the partner waves on the SIMD cover LDS latency; fragments are fetched per k-step to stay <= 128 VGPRs
#pragma unroll
      for (int kk = 0; kk < kSyrkKC / 4; ++kk) {
        double bv[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) bv[rb] = bw[(kk * 4 + l4) * kSyrkLd + (rb & 1) + 32 * (rb >> 1)];
#pragma unroll
        for (int half = 0; half < NCG / 4; ++half) {
          double av[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) av[q] = bl[(kk * 4 + l4) * kSyrkLd + (half * 4 + q) * 4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
              acc[half * 4 + q][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[q], bv[rb], acc[half * 4 + q][rb], 0, 0, 1 /* neg A */);
        }
      }
    }
  }
  if constexpr (DBG & 16) tk2 = wall_clock64();
  if (!active) return;
  if constexpr (DBG & 4) { double t = 0; for (int a = 0; a < NCG; ++a) for (int b = 0; b < 4; ++b) t += acc[a][b]; if (t == 1.2345e-300) F[0] = t; return; }
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    const int c = cbase + cg * 4 + l4;
    if (c >= climit) continue;
    double* colp = F + (size_t)c * f + cb_off(P, s, c, kfront);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = rbase + 2 * l15 + 32 * h;
      if (r + 1 < f && r >= c) {
        const d2_t v = (d2_t){acc[cg][2 * h], acc[cg][2 * h + 1]};
        __builtin_memcpy(colp + r, &v, 16);
      } else {
        if (r < f && r >= c) colp[r] = acc[cg][2 * h];
        if (r + 1 < f && r + 1 >= c) colp[r + 1] = acc[cg][2 * h + 1];
      }
    }
  }
  if constexpr (DBG & 16) {   // ticks of 10 ns: prologue + C load issue, main loop, store issue, store completion; per wave 0 of a workgroup
    tk3 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tk4 = wall_clock64();
    if (tid == 0) {
      unsigned long long* T = (unsigned long long*)(P.zero_page + 256);
      atomicAdd(&T[0], 1ull); atomicAdd(&T[1], (unsigned long long)(tk1 - tk0)); atomicAdd(&T[2], (unsigned long long)(tk2 - tk1));
      atomicAdd(&T[3], (unsigned long long)(tk3 - tk2)); atomicAdd(&T[4], (unsigned long long)(tk4 - tk3)); atomicAdd(&T[5], (unsigned long long)nchunk);
      atomicAdd(&T[6], (unsigned long long)(tkA - tk0)); if constexpr (DBG & 32) atomicAdd(&T[7], (unsigned long long)(tkB - tkA));
    }
  }
}

__global__ void k_set_shift(int n, const int* __restrict__ perm, double delta, int nshift, double* __restrict__ diagadd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) diagadd[i] = perm[i] < nshift ? delta : 0.0;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

template <typename T>
std::string upload(Numeric& N, const std::vector<T>& v, T** out) {
  *out = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  void* p = nullptr;
  OKKT_HIP_TRY(hipMalloc(&p, bytes));
  N.allocations.push_back(p);
  if (!v.empty()) OKKT_HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return "";
}

template <typename T>
std::string dalloc(Numeric& N, size_t count, T** out, bool zero) {
  *out = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  void* p = nullptr;
  OKKT_HIP_TRY(hipMalloc(&p, bytes));
  N.allocations.push_back(p);
  if (zero) {
    // hipMemset on device memory returns before the fill has run (null stream), and the handle's streams are
    // non-blocking: without this wait the first factorisation can race with the zero fill of its own buffers
    OKKT_HIP_TRY(hipMemset(p, 0, bytes));
    OKKT_HIP_TRY(hipStreamSynchronize(nullptr));
  } else if (getenv("OKKT_DEBUG_POISON")) {
    // debug: what is not zeroed (the front arena: the strict upper triangles of the fronts are never written) starts as NaNs instead of
    // whatever the allocator hands back -- a kernel that masks a stray read by multiplying with zero shows up (round 5: fuzz case 5000/126)
    OKKT_HIP_TRY(hipMemset(p, 0xFF, bytes));
    OKKT_HIP_TRY(hipStreamSynchronize(nullptr));
  }
  *out = (T*)p;
  return "";
}

size_t lds_small(int maxf) { return ((size_t)(maxf | 1) * maxf + maxf) * sizeof(double); }

}  // namespace

std::string numeric_setup(const Symbolic& S, const SymbolicOptions& opts, hipStream_t stream, Numeric& N) {
  N.stream = stream;
  N.nb = opts.panel_nb >= 128 ? 128 : (opts.panel_nb >= 64 ? 64 : 32);     // powers of two: kSolveBlock is a whole number of block columns
  N.small_max = std::max(32, std::min(opts.small_front_max, 136));
  // block columns per super-step (K = group * NB of the trailing update): 2 by default; fronts of at least
  // group_big_minf rows use group_big (more flops per byte of C traffic; the longer panel chain only pays off when the
  // trailing update is large).  OKKT_GROUP fixes one value for every front.
  N.group = 2;
  N.group_big = getenv("OKKT_GROUP_BIG") ? std::max(1, std::min(atoi(getenv("OKKT_GROUP_BIG")), 8)) : 4;
  N.group_big_minf = getenv("OKKT_GROUP_BIG_MINF") ? atoi(getenv("OKKT_GROUP_BIG_MINF")) : 8192;
  if (getenv("OKKT_GROUP")) { N.group = N.group_big = std::max(1, std::min(atoi(getenv("OKKT_GROUP")), 4)); }
  N.group_switch_rows = getenv("OKKT_GROUP_SWITCH_ROWS") ? atoi(getenv("OKKT_GROUP_SWITCH_ROWS")) : 9000;
  if (getenv("OKKT_SB_TAIL_ROWS")) N.sb_tail_rows = atoi(getenv("OKKT_SB_TAIL_ROWS"));   // -1: at the end of the factorisation only
  N.group_one_rows = getenv("OKKT_GROUP_ONE_ROWS") ? atoi(getenv("OKKT_GROUP_ONE_ROWS")) : 4000;
  N.nnz_in = S.nnz_in;
  if (S.max_front > 46000) return "front order exceeds the 32-bit local offset range";
  const int ns = S.nsuper;
  DevPlan& d = N.d;
  d.n = (int)S.n;
  d.nsuper = ns;
  d.has_dup = S.has_duplicates ? 1 : 0;
  std::string e;
#define UP(field, vec) if (!(e = upload(N, vec, &d.field)).empty()) return e
  UP(sn_col0, S.sn_col0);
  UP(row_ptr, S.row_ptr);
  UP(rows, S.rows);
  UP(child_ptr, S.child_ptr);
  UP(children, S.children);
  UP(rel_ptr, S.rel_ptr);
  UP(rel, S.rel);
  UP(cv_pos, S.cv_pos);
  UP(aent_ptr, S.aent_ptr);
  UP(aent_src, S.aent_src);
  UP(aent_dst, S.aent_dst);
  UP(perm, S.perm);
#undef UP
  // schedule: per level, fronts grouped by class, larger fronts first inside a class.  With a multi-GPU
  // partition there are two schedules: the subtrees this part owns and (part 0 only) the top of the tree.
  std::vector<int> sched;
  sched.reserve(ns);
  // W = L21 D of the big fronts' panels is dead once its level's launches have finished (the block inversions and the solves read F,
  // invl and xinv only), so a level's fronts take their W from one of TWO regions by the parity of the level: consecutive levels never
  // share one (the two-kernel form, now a patch, could still be in a level's last diagonal block when the next level started), and the
  // buffer is max over the even levels + max over the odd levels instead of the sum over all fronts (advisor, round 4: a second copy
  // of every pivot column, growing with nnz(L)).  wbuf_pos is relative to the region; the region's base is added below.
  std::vector<int64_t> wpos(ns, -1);
  std::vector<char> wparity(ns, 0);
  int64_t wregion[2] = {0, 0};
  N.n_small = N.n_big = 0;
  const bool parted = S.nparts > 1 && (int)S.sn_owner.size() == ns;
  // Tasks: a workgroup runs a whole subtree of small fronts, children before parents (the supernodes are numbered
  // in postorder, so a subtree is the index range [first descendant, root]); one launch per LEVEL OF TASKS instead of
  // one per level of fronts.  A banded KKT (the hanging chain of BASELINE config 2) has an elimination tree that is
  // one long path: 6668 levels of one 16-row front each = 6668 dependent launches (99 ms) without tasks, one launch
  // with them.  A subtree becomes one task when it is all small fronts and either little work (cost <= task_abs front
  // units) or without parallelism to lose (total cost <= task_ratio x its critical path); otherwise its root is a
  // task of its own and the rule is applied to the children.  Big fronts are units of their own.
  // (a partitioned plan: a task must not cross the cut -- only subtrees whose fronts all have the owner of their root become tasks;
  // round 3, until then partitioned plans ran one launch per level of fronts)
  static const bool parted_tasks = !(getenv("OKKT_PARTED_TASKS") && atoi(getenv("OKKT_PARTED_TASKS")) == 0);
  const bool use_tasks = (!parted || parted_tasks) && !(getenv("OKKT_TASKS") && atoi(getenv("OKKT_TASKS")) == 0);
  // plans of the level-structure dissection (banded systems) are all small fronts and nothing but launch latency: finer tasks
  // (12 front units, ratio 1.1) cut the hanging chain at the CUTEst size from 0.19 + 0.15 ms to 0.11 + 0.10 ms, N_h = 20 000 from
  // 0.37 + 0.29 to 0.24 + 0.19 ms (round 3 sweep); the general plans keep the coarser round-2 setting
  const bool level_nd = S.ordering_used == 4 && S.max_front <= N.small_max;   // the banded plans: all small fronts (a mesh-like system ordered by the level structure keeps the general setting)
  const double task_abs = getenv("OKKT_TASK_ABS") ? atof(getenv("OKKT_TASK_ABS")) : (level_nd ? 12.0 : 48.0);
  const double task_ratio = getenv("OKKT_TASK_RATIO") ? atof(getenv("OKKT_TASK_RATIO")) : (level_nd ? 1.1 : 1.5);
  std::vector<int> task_lo(ns), unit_root(ns), ulevel(ns, 0);
  auto fof = [&](int s2) { return (int)(S.row_ptr[s2 + 1] - S.row_ptr[s2]); };
  // forbid[s]: front s is a unit of its own (neither the root nor a member of a multi-front task); see the second pass below
  std::vector<char> forbid(ns, 0);
  auto form_units = [&]() {
    std::fill(ulevel.begin(), ulevel.end(), 0);
    std::vector<char> allsmall(ns), assigned(ns, 0);
    std::vector<int> first(ns);
    std::vector<double> ctot(ns), cpath(ns), cmaxchild(ns, 0.0);
    for (int s2 = 0; s2 < ns; ++s2) {
      const int f = fof(s2), k = S.sn_col0[s2 + 1] - S.sn_col0[s2];
      allsmall[s2] = f <= N.small_max && !forbid[s2];
      first[s2] = s2;
      ctot[s2] = 1.0 + (double)f * f * k / 8192.0;
      cpath[s2] = ctot[s2];
    }
    for (int s2 = 0; s2 < ns; ++s2) {           // children before parents
      cpath[s2] += cmaxchild[s2];
      const int p2 = S.sn_parent[s2];
      if (p2 < 0) continue;
      if (!allsmall[s2] || (parted && S.sn_owner[s2] != S.sn_owner[p2])) allsmall[p2] = 0;      // "all small" also means "all of one part"
      first[p2] = std::min(first[p2], first[s2]);
      ctot[p2] += ctot[s2];
      cmaxchild[p2] = std::max(cmaxchild[p2], cpath[s2]);
    }
    for (int s2 = ns - 1; s2 >= 0; --s2) {      // parents before children: the largest admissible subtrees win
      if (assigned[s2]) continue;
      task_lo[s2] = s2;
      unit_root[s2] = s2;
      assigned[s2] = 1;
      if (!use_tasks || !allsmall[s2] || !(ctot[s2] <= task_abs || ctot[s2] <= task_ratio * cpath[s2])) continue;
      task_lo[s2] = first[s2];
      for (int t = first[s2]; t < s2; ++t) { assigned[t] = 1; unit_root[t] = s2; task_lo[t] = t; }
    }
    for (int s2 = 0; s2 < ns; ++s2) {           // level of a unit = height in the tree of units
      const int p2 = S.sn_parent[s2];
      if (p2 < 0 || unit_root[p2] == unit_root[s2]) continue;
      ulevel[unit_root[p2]] = std::max(ulevel[unit_root[p2]], ulevel[unit_root[s2]] + 1);
    }
  };
  form_units();
  {
    // Second pass (round 6).  A level WITHOUT big fronts whose tasks are nearly all one-wave tasks (fronts of at most 32 rows) but for
    // a handful with a mid-size front ran that handful in launches of its own -- one 256-thread workgroup for 67 us of the
    // factorisation's critical path and 21 + 19 us of every solve at the metric size (53 869 one-wave tasks and ONE task with a
    // front of 65 .. 128 rows in level 0).  Those mid-size fronts become units of their own: they move up to the level above their
    // children, where the rule in build() below lets them join the big fronts of that level; what is left of their subtrees is
    // one-wave tasks like everything around it.
    static const int lone_max = getenv("OKKT_FOLD_LONE") ? atoi(getenv("OKKT_FOLD_LONE")) : 3;
    int nl = 0;
    for (int s2 = 0; s2 < ns; ++s2) if (unit_root[s2] == s2) nl = std::max(nl, ulevel[s2] + 1);
    std::vector<int> nbig(nl, 0), nmid(nl, 0), nwave(nl, 0);
    auto unit_maxf = [&](int s2) { int f = 0; for (int t = task_lo[s2]; t <= s2; ++t) f = std::max(f, fof(t)); return f; };
    for (int s2 = 0; s2 < ns; ++s2) {
      if (unit_root[s2] != s2) continue;
      const int f = unit_maxf(s2);
      ++(f <= 32 ? nwave : (f <= N.small_max ? nmid : nbig))[ulevel[s2]];
    }
    bool any = false;
    if (lone_max > 0 && use_tasks && !parted)
      for (int s2 = 0; s2 < ns; ++s2) {
        if (unit_root[s2] != s2) continue;
        const int l = ulevel[s2];
        if (nbig[l] != 0 || nmid[l] == 0 || nmid[l] > lone_max || nwave[l] < 64 || unit_maxf(s2) <= 32) continue;
        for (int t = task_lo[s2]; t <= s2; ++t) if (fof(t) > 32) { forbid[t] = 1; any = true; }
      }
    if (any) form_units();
  }
  int nulev = 0;
  for (int s2 = 0; s2 < ns; ++s2) if (unit_root[s2] == s2) nulev = std::max(nulev, ulevel[s2] + 1);
  std::vector<std::vector<int>> units_at(nulev);
  for (int s2 = 0; s2 < ns; ++s2) if (unit_root[s2] == s2) units_at[ulevel[s2]].push_back(s2);
  {
    // the latest level boundary that still has >= 30 % of the factorisation's flops above it (the root of S-metric: 81 %)
    std::vector<double> lev_flops(nulev, 0.0);
    double tot = 0.0;
    for (int s2 = 0; s2 < ns; ++s2) {
      const double f = (double)(S.row_ptr[s2 + 1] - S.row_ptr[s2]), k = (double)(S.sn_col0[s2 + 1] - S.sn_col0[s2]);
      const double fl = k * f * f - k * k * f + k * k * k / 3.0;
      lev_flops[ulevel[unit_root[s2]]] += fl;
      tot += fl;
    }
    N.early_level = -1;
    double above = 0.0;
    for (int l = nulev - 1; l >= 1; --l) {
      above += lev_flops[l];
      if (above >= 0.3 * tot) { N.early_level = l; break; }
    }
    if (parted) N.early_level = -1;
  }
  N.n_tasks = 0;
  N.max_task_len = 0;
  int wlevel_seq = 0;      // levels that hold big fronts, counted over both schedules (the top of a partitioned plan runs behind the local part)
  auto build = [&](std::vector<LevelSchedule>& levels, int want_owner) {
    levels.assign(nulev, LevelSchedule());
    for (int l = 0; l < nulev; ++l) {
      int64_t wlevel = 0;      // W doubles of this level's big fronts
      std::vector<int> cls[kNumClasses];
      for (int s : units_at[l]) {
        if (parted) { if (S.sn_owner[s] != want_owner) continue; }
        else if (want_owner != 0) continue;
        int f = 0;                                            // largest front of the unit decides the class
        for (int t = task_lo[s]; t <= s; ++t) f = std::max(f, (int)(S.row_ptr[t + 1] - S.row_ptr[t]));
        int c = f <= 32 ? 0 : (f <= 64 ? 1 : (f <= N.small_max ? 2 : 3));
        cls[c].push_back(s);
      }
      // A level that holds big fronts AND a handful of lone mid-size fronts (33 .. small_max rows, one front per unit) ran those in a
      // launch of their own in front of the big fronts' assembly, and in two more launches per solve: the metric workload has exactly
      // one such front -- 67 us of the factorisation's critical path and 21 + 19 us of every solve for one workgroup.  They join the big
      // fronts of their level instead (round-5 review, item 3a): a front of 33 .. 128 rows is one or two tiles of the dataflow launch and
      // a "thin" front of the solves.  OKKT_FOLD_LONE = how many such fronts a level may hold for the rule to apply (0: never).
      static const int fold_lone = getenv("OKKT_FOLD_LONE") ? atoi(getenv("OKKT_FOLD_LONE")) : 3;
      if (fold_lone > 0 && !cls[3].empty() && !cls[1].empty() + !cls[2].empty() > 0 && (int)(cls[1].size() + cls[2].size()) <= fold_lone) {
        for (int c = 1; c <= 2; ++c) {
          std::vector<int> keep;
          for (int s : cls[c]) (task_lo[s] == s ? cls[3] : keep).push_back(s);
          cls[c].swap(keep);
        }
      }
      if (getenv("OKKT_DEBUG_FRONTS")) fprintf(stderr, "okkt: level %d units by class: %zu (<= 32) %zu (<= 64) %zu (<= small_max) %zu (big)\n", l, cls[0].size(), cls[1].size(), cls[2].size(), cls[3].size());
      for (int c = 0; c < kNumClasses; ++c) {
        auto& v = cls[c];
        // longest tasks first (they finish last), then larger fronts first
        std::stable_sort(v.begin(), v.end(), [&](int a, int b) {
          const int la = a - task_lo[a], lb = b - task_lo[b];
          if (la != lb) return la > lb;
          return (S.row_ptr[a + 1] - S.row_ptr[a]) > (S.row_ptr[b + 1] - S.row_ptr[b]);
        });
        Segment& g = levels[l].seg[c];
        g.off = (int)sched.size();
        g.cnt = (int)v.size();
        for (int s : v) {
          for (int t = task_lo[s]; t <= s; ++t) {
            const int ft = (int)(S.row_ptr[t + 1] - S.row_ptr[t]);
            const int kt = S.sn_col0[t + 1] - S.sn_col0[t];
            g.maxf = std::max(g.maxf, ft);
            g.minf = std::min(g.minf, ft);
            g.maxk = std::max(g.maxk, kt);
            g.mink = std::min(g.mink, kt);
            if (c != 3) ++N.n_small;
          }
          const int f = (int)(S.row_ptr[s + 1] - S.row_ptr[s]);
          const int k = S.sn_col0[s + 1] - S.sn_col0[s];
          sched.push_back(s);
          if (c != 3) { ++N.n_tasks; N.max_task_len = std::max(N.max_task_len, s - task_lo[s] + 1); }
          if (c == 3 && getenv("OKKT_DEBUG_FRONTS")) fprintf(stderr, "okkt: big front level %d  f %d  k %d\n", (int)l, f, k);
          if (c == 3) {
            // two super-steps of W (look-ahead double buffer of the per-step schedule); the dataflow launch keeps the W of EVERY panel of
            // the front (a task may still read panel q while the chain is several block columns ahead: no slot is ever reused)
            const int64_t wcols = std::max<int64_t>((int64_t)N.nb * (f >= N.group_big_minf ? std::max(N.group, N.group_big) : N.group) * 2, N.dataflow ? ((int64_t)k + N.nb - 1) / N.nb * N.nb : 0);
            wpos[s] = wlevel; wparity[s] = (char)(wlevel_seq & 1); wlevel += (int64_t)f * wcols; ++N.n_big;
          }
        }
      }
      if (wlevel > 0) { wregion[wlevel_seq & 1] = std::max(wregion[wlevel_seq & 1], wlevel); ++wlevel_seq; }
    }
  };
  build(N.levels, parted ? N.part_id : 0);
  if (parted && N.part_id == 0) build(N.levels_top, -1); else N.levels_top.clear();
  if (!(e = upload(N, sched, &d.sched)).empty()) return e;
  if (!(e = upload(N, task_lo, &d.task_lo)).empty()) return e;
  // the leading levels without a big front run as one launch (k_front_small<.., FLOW>, the flow kernels of solve.hip)
  N.flow_levels = N.flow_off = N.flow_cnt = N.flow_maxf = N.flow_maxk = 0;
  if (N.flow && (!parted || parted_tasks)) {
    int nl = 0;
    while (nl < (int)N.levels.size() && N.levels[nl].seg[3].cnt == 0) ++nl;
    if (nl >= 2) {
      N.flow_levels = nl;
      N.flow_off = N.levels[0].seg[0].off;
      for (int l = 0; l < nl; ++l)
        for (int c = 0; c < 3; ++c) { N.flow_cnt += N.levels[l].seg[c].cnt; if (N.levels[l].seg[c].cnt) { N.flow_maxf = std::max(N.flow_maxf, N.levels[l].seg[c].maxf); N.flow_maxk = std::max(N.flow_maxk, N.levels[l].seg[c].maxk); } }
      if (N.flow_cnt == 0) N.flow_levels = 0;
    }
  }
  {
    // backward sweep of those levels: a task waits for the task that holds its root's parent front -- if that one is part of the
    // same launch (a parent above the flow levels has been solved by an earlier launch)
    std::vector<int> unit_parent(ns, -1);
    for (int s2 = 0; s2 < ns; ++s2) {
      if (unit_root[s2] != s2 || S.sn_parent[s2] < 0) continue;
      const int up = unit_root[S.sn_parent[s2]];
      if (ulevel[up] < N.flow_levels && (!parted || S.sn_owner[up] == S.sn_owner[s2])) unit_parent[s2] = up;     // a parent in the top schedule of a partitioned plan is not part of the launch
    }
    if (!(e = upload(N, unit_parent, &d.unit_parent)).empty()) return e;
  }
  if (getenv("OKKT_DEBUG_FRONTS")) fprintf(stderr, "okkt: %d levels of units (%d levels of fronts), %lld tasks of small fronts, longest %d\n", nulev, S.nlevels, (long long)N.n_tasks, N.max_task_len);
  // ---- front arena (round 6: contribution blocks released).  Rounds 1 - 5 gave every front a dense f x f buffer for the plan's lifetime
  // ("sized for 288 GB"): 11.8 GB for the 0.86 GB factor of the metric workload, so that a ten times larger system of the same family
  // did not fit one GPU.  Only the L panel (f x k) has to stay -- the solves read it; the r x r contribution block is dead once the parent
  // front has been assembled.  The panels are laid out one after the other; a contribution block (stored as the r trailing columns of the
  // front WITH the front's leading dimension, so that every kernel keeps addressing column c at F + c f) gets a place in a shared region
  // from a lifetime-aware first-fit allocator run over the schedule: alive from the epoch of its front to the epoch of its parent --
  // an epoch being one level of units, except that the leading levels that run as ONE launch (flow levels) are one epoch: a task of
  // that launch starts as soon as its own children are done, whatever the rest of its level is doing.
  // Partitioned plans keep the f x f buffers (boundary blocks cross the local / top schedules and the exchange kernels).
  N.release_cb = getenv("OKKT_RELEASE_CB") ? atoi(getenv("OKKT_RELEASE_CB")) : 1;
  if (parted) N.release_cb = 0;
  std::vector<int64_t> fpos(ns + 1, 0), cbshift(ns, 0);
  if (!N.release_cb) {
    for (int s2 = 0; s2 <= ns; ++s2) fpos[s2] = S.front_pos[s2];
    N.arena_doubles = S.arena_doubles;
    N.cb_region_doubles = 0;
  } else {
    for (int s2 = 0; s2 < ns; ++s2) {
      const int64_t f = S.row_ptr[s2 + 1] - S.row_ptr[s2], k = S.sn_col0[s2 + 1] - S.sn_col0[s2];
      fpos[s2 + 1] = fpos[s2] + ((f * k + 1) & ~(int64_t)1);      // even: 16-byte pairs stay aligned where they were
    }
    const int64_t panels = fpos[ns];
    auto epoch_of = [&](int s2) { const int l = ulevel[unit_root[s2]]; return l < N.flow_levels ? 0 : l - std::max(N.flow_levels, 1) + 1; };
    const int nep = std::max(1, nulev - std::max(N.flow_levels, 1) + 1) + 1;
    std::vector<std::vector<int>> born(nep), dies(nep);
    for (int s2 = 0; s2 < ns; ++s2) {
      const int64_t f = S.row_ptr[s2 + 1] - S.row_ptr[s2], k = S.sn_col0[s2 + 1] - S.sn_col0[s2];
      if (f == k) continue;
      const int p2 = S.sn_parent[s2];
      born[epoch_of(s2)].push_back(s2);
      dies[p2 >= 0 ? epoch_of(p2) : nep - 1].push_back(s2);
    }
    // free list: offset -> size, coalesced; best fit
    std::map<int64_t, int64_t> freeb;
    int64_t top = 0;
    std::vector<int64_t> cbo(ns, -1), cbsz(ns, 0);
    auto release = [&](int64_t off, int64_t sz) {
      auto it = freeb.emplace(off, sz).first;
      auto nx = std::next(it);
      if (nx != freeb.end() && it->first + it->second == nx->first) { it->second += nx->second; freeb.erase(nx); }
      if (it != freeb.begin()) { auto pv = std::prev(it); if (pv->first + pv->second == it->first) { pv->second += it->second; freeb.erase(it); it = pv; } }
      if (it->first + it->second == top) { top = it->first; freeb.erase(it); }
    };
    for (int ep = 0; ep < nep; ++ep) {
      std::vector<int>& v = born[ep];
      std::stable_sort(v.begin(), v.end(), [&](int a, int b) {
        const int64_t fa = S.row_ptr[a + 1] - S.row_ptr[a], fb = S.row_ptr[b + 1] - S.row_ptr[b];
        const int64_t sa = fa * (fa - (S.sn_col0[a + 1] - S.sn_col0[a])), sb = fb * (fb - (S.sn_col0[b + 1] - S.sn_col0[b]));
        return sa > sb;
      });
      for (int s2 : v) {
        const int64_t f = S.row_ptr[s2 + 1] - S.row_ptr[s2], k = S.sn_col0[s2 + 1] - S.sn_col0[s2];
        const int64_t sz = ((f - k) * f + 1) & ~(int64_t)1;
        auto best = freeb.end();
        for (auto it = freeb.begin(); it != freeb.end(); ++it)
          if (it->second >= sz && (best == freeb.end() || it->second < best->second)) best = it;
        int64_t off;
        if (best != freeb.end()) {
          off = best->first;
          const int64_t rest = best->second - sz;
          freeb.erase(best);
          if (rest > 0) freeb.emplace(off + sz, rest);
        } else { off = top; top += sz; }
        cbo[s2] = off; cbsz[s2] = sz;
      }
      for (int s2 : dies[ep]) if (cbo[s2] >= 0) release(cbo[s2], cbsz[s2]);
    }
    int64_t region = 0;
    for (int s2 = 0; s2 < ns; ++s2) if (cbo[s2] >= 0) region = std::max(region, cbo[s2] + cbsz[s2]);
    for (int s2 = 0; s2 < ns; ++s2) {
      if (cbo[s2] < 0) continue;
      const int64_t f = S.row_ptr[s2 + 1] - S.row_ptr[s2], k = S.sn_col0[s2 + 1] - S.sn_col0[s2];
      cbshift[s2] = (panels + cbo[s2]) - (fpos[s2] + k * f);
    }
    N.arena_doubles = panels + region;
    N.cb_region_doubles = region;
    if (getenv("OKKT_DEBUG_FRONTS")) fprintf(stderr, "okkt: front arena %.3f GB (panels %.3f, shared contribution-block region %.3f) instead of %.3f GB of f x f buffers\n",
                                            N.arena_doubles * 8e-9, panels * 8e-9, region * 8e-9, S.arena_doubles * 8e-9);
  }
  N.front_pos_host = fpos;
  if (!(e = upload(N, fpos, &d.front_pos)).empty()) return e;
  if (!(e = upload(N, cbshift, &d.cb_shift)).empty()) return e;
  for (int s2 = 0; s2 < ns; ++s2) if (wpos[s2] >= 0 && wparity[s2]) wpos[s2] += wregion[0];
  const int64_t wtotal = wregion[0] + wregion[1];
  if (!(e = upload(N, wpos, &d.wbuf_pos)).empty()) return e;
  N.sched_host = sched;
  {
    std::vector<int> owner(ns, 0), col_owner((size_t)S.n, 0);
    if (parted) owner = S.sn_owner;
    for (int sn = 0; sn < ns; ++sn)
      for (int j = S.sn_col0[sn]; j < S.sn_col0[sn + 1]; ++j) col_owner[j] = owner[sn];
    if (!(e = upload(N, owner, &d.sn_owner)).empty()) return e;
    if (!(e = upload(N, col_owner, &d.col_owner)).empty()) return e;
    {
      std::vector<int> top_cols;
      if (parted) for (int j = 0; j < (int)S.n; ++j) if (col_owner[j] == -1) top_cols.push_back(j);
      N.n_top_cols = (int)top_cols.size();
      if (!(e = upload(N, top_cols, &d.top_cols)).empty()) return e;
    }
    N.n_boundary = parted ? (int)S.boundary.size() : 0;
    if (parted) {
      if (!(e = upload(N, S.boundary, &d.bnd)).empty()) return e;
      if (!(e = upload(N, S.boundary_cb, &d.bnd_cb)).empty()) return e;
      if (!(e = upload(N, S.boundary_cv, &d.bnd_cv)).empty()) return e;
    }
  }
  N.sn_f.resize(ns);
  N.sn_k.resize(ns);
  for (int s = 0; s < ns; ++s) { N.sn_f[s] = (int)(S.row_ptr[s + 1] - S.row_ptr[s]); N.sn_k[s] = S.sn_col0[s + 1] - S.sn_col0[s]; }
  // big fronts: inverted extend-add lists (per front column: which (child, jj) land on it),
  // storage for the inverse diagonal blocks and the forward-solve work vectors
  {
    std::vector<int64_t> bigcol_base(ns, -1), invl_pos(ns, -1);
    int64_t nbigcols = 0, invl_total = 0;
    for (int s = 0; s < ns; ++s)
      if (wpos[s] >= 0) {
        const int64_t f = S.row_ptr[s + 1] - S.row_ptr[s];
        const int64_t k = S.sn_col0[s + 1] - S.sn_col0[s];
        bigcol_base[s] = nbigcols;
        nbigcols += f;
        invl_pos[s] = invl_total;
        invl_total += ((k + N.nb - 1) / N.nb) * (int64_t)N.nb * N.nb;
      }
    std::vector<int64_t> ea_ptr(nbigcols + 1, 0);
    for (int c = 0; c < ns; ++c) {
      const int p = S.sn_parent[c];
      if (p < 0 || wpos[p] < 0) continue;
      for (int64_t q = S.rel_ptr[c]; q < S.rel_ptr[c + 1]; ++q) ++ea_ptr[bigcol_base[p] + S.rel[q] + 1];
    }
    for (int64_t i = 0; i < nbigcols; ++i) ea_ptr[i + 1] += ea_ptr[i];
    std::vector<int> ea_child(ea_ptr[nbigcols]), ea_jj(ea_ptr[nbigcols]), ea_rc(ea_ptr[nbigcols]);
    std::vector<int64_t> ea_src(ea_ptr[nbigcols]), ea_rel(ea_ptr[nbigcols]), ea_cut(ea_ptr[nbigcols]), ea_pos(ea_ptr[nbigcols]);
    // per child of a big front: where each kAsmChunk-row boundary of the parent falls in its (sorted) rel list
    std::vector<int64_t> cut_pos(ns, -1);
    std::vector<int> cutv;
    for (int c = 0; c < ns; ++c) {
      const int p = S.sn_parent[c];
      if (p < 0 || wpos[p] < 0) continue;
      const int64_t fp = S.row_ptr[p + 1] - S.row_ptr[p];
      const int* rb = S.rel.data() + S.rel_ptr[c];
      const int* re = S.rel.data() + S.rel_ptr[c + 1];
      cut_pos[c] = (int64_t)cutv.size();
      const int nchunks = (int)((fp + kAsmChunk - 1) / kAsmChunk);
      for (int t = 0; t <= nchunks; ++t) cutv.push_back((int)(std::lower_bound(rb, re, t * kAsmChunk) - rb));
    }
    // per big-front column: first A entry of the column (entries are sorted by destination)
    std::vector<int64_t> acol_lo(nbigcols + 1, 0);
    for (int s = 0; s < ns; ++s)
      if (wpos[s] >= 0) {
        const int64_t f = S.row_ptr[s + 1] - S.row_ptr[s];
        const int64_t e0 = S.aent_ptr[s], e1 = S.aent_ptr[s + 1];
        int64_t e = e0;
        for (int64_t pc = 0; pc < f; ++pc) {
          while (e < e1 && (int64_t)S.aent_dst[e] < pc * f) ++e;
          acol_lo[bigcol_base[s] + pc] = e;
        }
      }
    std::vector<int64_t> fill(ea_ptr.begin(), ea_ptr.end() - 1);
    for (int c = 0; c < ns; ++c) {  // children in ascending order: the summation order is fixed
      const int p = S.sn_parent[c];
      if (p < 0 || wpos[p] < 0) continue;
      for (int64_t q = S.rel_ptr[c]; q < S.rel_ptr[c + 1]; ++q) {
        const int64_t slot = fill[bigcol_base[p] + S.rel[q]]++;
        ea_child[slot] = c;
        ea_jj[slot] = (int)(q - S.rel_ptr[c]);
        ea_pos[slot] = S.cv_pos[c] + (q - S.rel_ptr[c]);
        {
          const int64_t kc = S.sn_col0[c + 1] - S.sn_col0[c], fc = S.row_ptr[c + 1] - S.row_ptr[c];
          ea_rc[slot] = (int)(fc - kc);
          ea_rel[slot] = S.rel_ptr[c];
          ea_cut[slot] = cut_pos[c];
          ea_src[slot] = fpos[c] + cbshift[c] + (kc + ea_jj[slot]) * fc + kc;   // arena offset of the child's CB column, row 0 of the CB
        }
      }
    }
    if (!(e = upload(N, bigcol_base, &d.bigcol_base)).empty()) return e;
    if (!(e = upload(N, invl_pos, &d.invl_pos)).empty()) return e;
    if (!(e = upload(N, ea_ptr, &d.ea_ptr)).empty()) return e;
    if (!(e = upload(N, ea_child, &d.ea_child)).empty()) return e;
    if (!(e = upload(N, ea_jj, &d.ea_jj)).empty()) return e;
    if (!(e = upload(N, ea_pos, &d.ea_pos)).empty()) return e;
    if (!(e = upload(N, ea_rc, &d.ea_rc)).empty()) return e;
    if (!(e = upload(N, ea_src, &d.ea_src)).empty()) return e;
    if (!(e = upload(N, ea_rel, &d.ea_rel)).empty()) return e;
    if (!(e = upload(N, ea_cut, &d.ea_cut)).empty()) return e;
    {
      std::vector<EaRec> recs(ea_src.size());
      for (size_t t = 0; t < recs.size(); ++t) recs[t] = EaRec{ea_src[t], ea_rel[t], ea_rc[t], ea_jj[t], ea_cut[t]};
      EaRec* dr = nullptr;
      if (!(e = upload(N, recs, &dr)).empty()) return e;
      d.ea_rec = dr;
    }
    if (!(e = upload(N, cutv, &d.cutv)).empty()) return e;
    if (!(e = upload(N, acol_lo, &d.acol_lo)).empty()) return e;
    // zero-filled once: the parts of a block beyond a front's last pivot column are never written and are read as zeros
    if (!(e = dalloc(N, (size_t)invl_total, &d.invl, true)).empty()) return e;
  }
  // (+ a tail of slack: loads on clamped addresses stay inside the allocation whatever a masked lane computes)
  if (!(e = dalloc(N, (size_t)N.arena_doubles + 512 + (size_t)S.max_front * 2, &d.arena, false)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.n, &d.dvals, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.n, &d.diagadd, true)).empty()) return e;
  // kMaxRhs right-hand sides travel through the sweeps together (solve.hip)
  d.xw_stride = S.n;
  d.cv_stride = S.sum_r;
  if (!(e = dalloc(N, (size_t)S.n * kMaxRhs, &d.xwork, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.n * kMaxRhs, &d.zwork, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.sum_r * kMaxRhs, &d.cv, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)wtotal + 512, &d.wbuf, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)kCountSlots * kCountStride, &d.counters, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)512, &d.zero_page, true)).empty()) return e;   // [256, 512): debug tick counters (OKKT_DEBUG_SYRK=96)
  if (!(e = dalloc(N, (size_t)S.nnz_in, &N.vals_owned, false)).empty()) return e;
  {
    double* raw = nullptr;       // one monotonic flag (int) per supernode for the fused diag + trsm launches
    if (!(e = dalloc(N, (size_t)ns / 2 + 8, &raw, true)).empty()) return e;
    N.chain_flags = (int*)raw;
    N.chain_epoch = 0;
    raw = nullptr;               // three per supernode for the multi-level launches of small-front tasks
    if (!(e = dalloc(N, (size_t)3 * ns / 2 + 8, &raw, true)).empty()) return e;
    N.flow_flags = (int*)raw;
    N.flow_epoch = 0;
  }
  if (!(e = solve_setup(S, N)).empty()) return e;
  if (!(e = df_setup(N)).empty()) return e;
  // kernels that may want more than 64 KiB of dynamic LDS
  const int big_lds = 160 * 1024 - 64;   // the stop-flag check keeps one static LDS word per kernel
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_front_small<256>, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_front_small<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_big_diag, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_big_diag2, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_diag_trsm_fused, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_big_invert, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_big_trsm<3>, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_big_trsm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  for (const void* fn : {(const void*)k_big_syrk<0, kSyrkTrail>, (const void*)k_big_syrk<0, kSyrkTrail, 64>, (const void*)k_big_syrk<0, kSyrkPanel>, (const void*)k_big_syrk<0, kSyrkAhead>,
                         (const void*)k_big_syrk<1, kSyrkTrail>, (const void*)k_big_syrk<2, kSyrkTrail>, (const void*)k_big_syrk<5, kSyrkTrail>,
                         (const void*)k_big_syrk<13, kSyrkTrail>, (const void*)k_big_syrk<16, kSyrkTrail>, (const void*)k_big_syrk<16, kSyrkTrail, 64>, (const void*)k_big_syrk<48, kSyrkTrail>})
    OKKT_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  return "";
}

void numeric_release(Numeric& N) {
  for (hipEvent_t ev : N.prof_events) (void)hipEventDestroy(ev);
  N.prof_events.clear();
  N.prof_used = 0;
  N.prof_flops.clear();
  if (N.inv_event) { (void)hipEventDestroy(N.inv_event); N.inv_event = nullptr; }
  for (hipEvent_t ev : N.inv_level_events) if (ev) (void)hipEventDestroy(ev);
  N.inv_level_events.clear(); N.inv_level_pending.clear();
  N.inv_wait = false;
  for (hipEvent_t ev : N.la_events) (void)hipEventDestroy(ev);
  N.la_events.clear();
  N.la_used = 0;
  for (void* p : N.allocations) (void)hipFree(p);
  N.allocations.clear();
  N.levels.clear();
  N.levels_top.clear();
  N.chain_flags = nullptr; N.chain_epoch = 0;
  N.df_tasks = nullptr; N.df_heads = nullptr; N.n_df_heads = 0; N.df_state_ints = 0;
  N.flow_flags = nullptr; N.flow_epoch = 0; N.flow_levels = 0;
  N.solve_flags = nullptr; N.solve_counters = nullptr; N.solve_epoch = 0; N.solve_counters64 = nullptr; N.solve_epoch64 = 0;
  N.slevels.clear();
  N.slevels_top.clear();
  N.d = DevPlan();
  N.vals_owned = nullptr;
}

std::string numeric_read_counts(Numeric& N, hipStream_t stream, unsigned long long out[6]) {
  unsigned long long raw[kCountSlots * kCountStride];
  OKKT_HIP_TRY(hipMemcpyAsync(raw, N.d.counters, sizeof(raw), hipMemcpyDeviceToHost, stream));
  OKKT_HIP_TRY(hipStreamSynchronize(stream));
  for (int c = 0; c < 4; ++c) {
    out[c] = 0;
    for (int q = 0; q < kCountSlots; ++q) out[c] += raw[q * kCountStride + c];
  }
  out[4] = raw[4];
  out[5] = raw[5];
  return "";
}

// one schedule (the local subtrees of a part, or the top of the tree) on the streams of `ss`
static std::string factor_sched(Numeric& N, DevPlan P, const std::vector<LevelSchedule>& levels, const std::vector<SolveLevel>& slevels,
                                const LaneStreams& ss, double tol, bool in_loop_check, bool& inv_on_aux, size_t l_begin = 0, size_t l_end = (size_t)-1);

// the words a factorisation starts from zero, in ONE launch: three hipMemsetAsync are three fill kernels of 4 - 5 us each, 14 of the 108 us
// a CUTEst-size system (BASELINE config 2) takes to factor (round 6, profiles/r06_small_configs_trace.txt)
__global__ __launch_bounds__(256) void k_zero_words(unsigned* __restrict__ a, size_t na, unsigned* __restrict__ b, size_t nb, unsigned* __restrict__ c, size_t nc) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb + nc; i += stride) {
    if (i < na) a[i] = 0u;
    else if (i < na + nb) b[i - na] = 0u;
    else c[i - na - nb] = 0u;
  }
}

std::string numeric_factor_enqueue(Numeric& N, const double* d_vals, double tol, int which, bool reset_counters) {
  DevPlan P = N.d;
  P.vals = d_vals;
  hipStream_t st = N.stream;
  {
    // the pivot counters; the arrival counters of the fused backward launches (reset by their consumer, but a sweep that was cut short -- a
    // wait that ran into its bound -- leaves a residue that would let the next solve's consumer start early: cleared with the time-out word,
    // advisor, round 3); the tile states and queue heads of the dataflow launches
    const size_t na = reset_counters ? (size_t)kCountSlots * kCountStride * (sizeof(unsigned long long) / sizeof(unsigned)) : 0;
    const size_t nb = (reset_counters && N.solve_counters) ? (size_t)N.d.nsuper : 0;
    const size_t nc = (N.dataflow && P.df_state) ? (size_t)N.df_state_ints + (size_t)N.n_df_heads * kDfHeadStride : 0;
    if (na + nb + nc) {
      const unsigned grid = (unsigned)std::min<size_t>(2048, (na + nb + nc + 1023) / 1024);
      hipLaunchKernelGGL(k_zero_words, dim3(grid), dim3(256), 0, st, (unsigned*)P.counters, na, (unsigned*)N.solve_counters, nb, (unsigned*)P.df_state, nc);
    }
  }
  if (N.early_check && N.early_device && which == 0 && N.levels_top.empty() && reset_counters) { P.want_pos = N.early_n; P.want_neg = N.early_m; }
  N.la_used = 0;
  // A factorisation right behind another one (a retry of the delta loop: no solve in between): the previous factor's block inversions may
  // still be running on the auxiliary stream and read the arena and invl that the new assembly overwrites -- the handle's stream waits for
  // them (no cost when a solve ran in between: it has waited already)
  if (which == 0) {
    for (size_t l = 0; l < N.inv_level_pending.size(); ++l)
      if (N.inv_level_pending[l] && N.inv_level_events[l]) OKKT_HIP_TRY(hipStreamWaitEvent(st, N.inv_level_events[l], 0));
    if (N.inv_wait && N.inv_event) OKKT_HIP_TRY(hipStreamWaitEvent(st, N.inv_event, 0));
  }
  if (which == 0) std::fill(N.inv_level_pending.begin(), N.inv_level_pending.end(), 0);
  if (which == 0) N.inv_wait = false;   // the block inverses belong to the previous factorisation (the top phase of a partitioned
                                        // plan keeps the wait its local phase has set up)
  bool inv_on_aux = false;
  N.early_exited = false;
  std::string e;
  {
    const LaneStreams ss{N.stream, N.stream_masked, N.stream_panel, N.stream_aux};
    e = factor_sched(N, P, which == 0 ? N.levels : N.levels_top, which == 0 ? N.slevels : N.slevels_top, ss, tol,
                     which == 0 && N.levels_top.empty(), inv_on_aux);
    if (!e.empty()) return e;
  }
  const bool per_level = which == 0 && N.levels_top.empty() && !N.inv_level_pending.empty();
  if (inv_on_aux && !per_level) {    // the next solve waits for the block inversions that are still running on the auxiliary stream
    if (!N.inv_event) OKKT_HIP_TRY(hipEventCreateWithFlags(&N.inv_event, hipEventDisableTiming));
    OKKT_HIP_TRY(hipEventRecord(N.inv_event, N.inv_stream ? N.inv_stream : N.stream_aux));
    N.inv_wait = true;
  }
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}

static std::string factor_sched(Numeric& N, DevPlan P, const std::vector<LevelSchedule>& levels, const std::vector<SolveLevel>& slevels,
                                const LaneStreams& ss, double tol, bool in_loop_check, bool& inv_on_aux, size_t l_begin, size_t l_end) {
  const hipStream_t cur = ss.main;      // the stream the schedule is on
  const int NB = N.nb;
  static const int dbg_syrk = getenv("OKKT_DEBUG_SYRK") ? atoi(getenv("OKKT_DEBUG_SYRK")) : 0;
  static const int split_min_rows = getenv("OKKT_SPLIT_MIN_ROWS") ? atoi(getenv("OKKT_SPLIT_MIN_ROWS")) : 5000;
  static const int dbg_stop = getenv("OKKT_DEBUG_DIAG_STOP") ? atoi(getenv("OKKT_DEBUG_DIAG_STOP")) : 0;
  static const int fuse_max_fronts = getenv("OKKT_FUSE_MAX_FRONTS") ? atoi(getenv("OKKT_FUSE_MAX_FRONTS")) : 8;
  for (size_t l = l_begin; l < std::min(l_end, levels.size()); ++l) {
    const LevelSchedule& L = levels[l];
    if (N.early_check && in_loop_check && (int)l == N.early_level) {
      // the pivots counted so far already decide a wrong inertia?  Then the (expensive) rest of the tree is skipped:
      // one synchronisation per factorisation, 35 of 49 ms saved per failed attempt of the delta loop at S-metric
      unsigned long long cnt[6] = {0, 0, 0, 0, 0, 0};
      std::string ec = numeric_read_counts(N, cur, cnt);
      if (!ec.empty()) return ec;
      if (cnt[4] != 0 || cnt[3] > 0 || cnt[2] > 0 || cnt[1] > (unsigned long long)N.early_m || cnt[0] > (unsigned long long)N.early_n) {
        N.early_exited = true;
        return "";
      }
    }
    if (l == 0 && &levels == &N.levels && N.flow_levels >= 2 && l_end >= (size_t)N.flow_levels && N.flow_flags) {
      // levels [0, flow_levels): every task in one launch, children tasks at lower block indices than their parents
      const int ep = ++N.flow_epoch;
      // OKKT_DEBUG_DROP_HANDOFF=1 (tests): the consumers wait for an epoch nobody raises -- every wait runs into its bound
      static const int drop = getenv("OKKT_DEBUG_DROP_HANDOFF") ? atoi(getenv("OKKT_DEBUG_DROP_HANDOFF")) : 0;
      const int wep = ep + ((drop & 1) ? (1 << 20) : 0);
      if (N.flow_maxf <= 32) hipLaunchKernelGGL((k_front_small<64, true>), dim3(N.flow_cnt), dim3(64), lds_small(N.flow_maxf), cur, P, P.sched + N.flow_off, tol, N.flow_flags, ep, wep);
      else hipLaunchKernelGGL((k_front_small<256, true>), dim3(N.flow_cnt), dim3(256), lds_small(N.flow_maxf), cur, P, P.sched + N.flow_off, tol, N.flow_flags, ep, wep);
      if (P.want_neg >= 0) hipLaunchKernelGGL(k_fold_counts, dim3(1), dim3(64), 0, cur, P);
      l = (size_t)N.flow_levels - 1;
      continue;
    }
    if (L.seg[0].cnt) {
      const Segment& g = L.seg[0];
      hipLaunchKernelGGL(k_front_small<64>, dim3(g.cnt), dim3(64), lds_small(g.maxf), cur, P, P.sched + g.off, tol, (int*)nullptr, 0, 0);
    }
    for (int c = 1; c <= 2; ++c)
      if (L.seg[c].cnt) {
        const Segment& g = L.seg[c];
        hipLaunchKernelGGL(k_front_small<256>, dim3(g.cnt), dim3(256), lds_small(g.maxf), cur, P, P.sched + g.off, tol, (int*)nullptr, 0, 0);
      }
    if (P.want_neg >= 0 && (L.seg[0].cnt || L.seg[1].cnt || L.seg[2].cnt)) hipLaunchKernelGGL(k_fold_counts, dim3(1), dim3(64), 0, cur, P);
    if (L.seg[3].cnt) {
      const Segment& g = L.seg[3];
      const int* list = P.sched + g.off;
      const int nsteps = (g.maxk + NB - 1) / NB;
      const size_t lds_diag = ((size_t)3 * kMW * kPLD + (size_t)2 * 4 * 32 * kXld) * sizeof(double);
      const size_t lds_diag2 = OKKT_DIAG2_LDS_DOUBLES(kMW) * sizeof(double);
      const size_t lds_trsm_max = ((size_t)10 * kIB * kIB + 128) * sizeof(double);
      // every front of the segment must own a W buffer for this group size: decided on the smallest front
      const int GS = g.minf >= N.group_big_minf ? N.group_big : N.group;     // widest super-step of this segment (sizes wbuf)
      // Super-step widths adapt: GS block columns while the trailing update is long enough to hide the panel chain,
      // N.group (2) once the chain is the critical path -- there the in-group diagonal-tile updates with K up to
      // (GS - 1) * NB (33-75 us each for a lone tile) cost more than the extra C traffic of K = 256 updates.
      auto gs_at = [&](int stepA) {
        const int rows_after = g.maxf - (stepA + GS) * NB;
        if (rows_after < N.group_one_rows) return 1;
        return (GS > N.group && rows_after < N.group_switch_rows) ? N.group : GS;
      };
      auto next_event = [&](hipEvent_t* ev) -> std::string {
        if (N.la_used >= N.la_events.size())
          for (int q = 0; q < 64; ++q) { hipEvent_t e2; OKKT_HIP_TRY(hipEventCreateWithFlags(&e2, hipEventDisableTiming)); N.la_events.push_back(e2); }
        *ev = N.la_events[N.la_used++];
        return "";
      };
      // look-ahead for the trailing update of the super-step that starts at block column stepA (gs_cur wide)?
      auto la_at = [&](int stepA, int gs_cur) {
        const int stepB = stepA + gs_cur;
        if (stepB >= nsteps || !N.lookahead || ss.panel == nullptr || ss.masked == nullptr) return false;
        const int remr = g.maxf - (stepB + gs_at(stepB)) * NB;     // rows of the rest triangle (upper bound)
        const int Trr = remr > 0 ? (remr + 127) / 128 : 0;
        return (int64_t)Trr * (Trr + 1) / 2 * g.cnt >= N.la_min_tiles;
      };
      // A level that uses the look-ahead runs on the CU-masked twin of the handle's stream (the reserved CUs belong
      // to the panel streams); every other level keeps all CUs
      // the level's big fronts as one persistent dataflow launch (dataflow.hip) instead of the per-step launches below
      const bool use_df = N.dataflow && g.df_cnt > 0 && NB == 128 && dbg_syrk == 0 && dbg_stop == 0 && P.df_state != nullptr;
      const bool seg_la = !use_df && la_at(0, gs_at(0));
      hipStream_t st = cur;
      if (seg_la) {
        hipEvent_t evf;
        std::string e = next_event(&evf);
        if (!e.empty()) return e;
        OKKT_HIP_TRY(hipEventRecord(evf, ss.main));
        OKKT_HIP_TRY(hipStreamWaitEvent(ss.masked, evf, 0));
        st = ss.masked;
      }
      {
        // LDS-resident columns up to 2048 rows (16 KiB per wave); smaller fronts take less LDS for more waves per CU
        static const int lcol_max = getenv("OKKT_ASM_LCOL") ? atoi(getenv("OKKT_ASM_LCOL")) : 2048;
        // (the level's longest column rounded up to 64 rows, not to a power of two: 1 100 rows are 36 KiB per workgroup and four
        // workgroups per CU where 2 048 were 64 KiB and two -- the kernel hides the latency of its dependent record fetches by waves:
        // S-metric 16.78 -> 16.60 ms, S-C5 3.91 -> 3.83.  The later, shorter columns in launches of their own with still less LDS:
        // measured, slower -- every extra launch of a level is a serial bubble, S-C5 3.83 -> 3.89 / 3.92 / 3.97 ms with 2 / 3 / 4 ranges)
        int lcol = std::min(lcol_max, std::max(256, (g.maxf + 63) / 64 * 64));
        if (lcol_max <= 0) lcol = 0;
        // lower levels (many fronts, short items): loads of four items batched; upper levels: item after item
        static const int chunked = getenv("OKKT_ASM_CHUNKED") ? atoi(getenv("OKKT_ASM_CHUNKED")) : 1;
        static const int chunk_min = getenv("OKKT_ASM_CHUNK_MIN") ? atoi(getenv("OKKT_ASM_CHUNK_MIN")) : 2048;
        if (g.maxf > chunk_min && chunked) hipLaunchKernelGGL(k_big_assemble_chunked, dim3((g.maxf + 3) / 4, g.cnt, (g.maxf + kAsmChunk - 1) / kAsmChunk), dim3(256), 0, st, P, list);
        else if (g.maxf <= 2048) hipLaunchKernelGGL(k_big_assemble<true>, dim3((g.maxf + 3) / 4, g.cnt), dim3(256), (size_t)4 * lcol * sizeof(double), st, P, list, lcol);
        else hipLaunchKernelGGL(k_big_assemble<false>, dim3((g.maxf + 3) / 4, g.cnt), dim3(256), (size_t)4 * lcol * sizeof(double), st, P, list, lcol);
      }
      // W of the super-step with parity `par` lives in wbuf columns [par * GS * NB, ...): the look-ahead factors the
      // panels of the next super-step while the trailing update of the current one still reads its W
      // npan panels [stepA, stepA + npan) (their W at parity par) applied to the region from block column tstep; cs = block
      // columns of the NEXT super-step (look-ahead split of the region)
      auto launch_syrk = [&](hipStream_t sst, int stepA, int npan, int tstep, int head, int par, int cs, int sub = 0) -> std::string {
        // upper bound on the rows of the target region: a front whose pivot block ends inside the group
        // starts its trailing region at k < tstep * NB
        const int rem = g.maxf - (head == 1 ? tstep : stepA) * NB;
        if (rem <= 0) return "";
        const int T = (rem + 127) / 128;
        const int Tr = std::max(T - cs, 0);
        int ntile = T * (T + 1) / 2;
        if (head == 1) ntile = sub == 1 ? 1 : (sub == 2 ? T - 1 : T);
        else if (head == 2) { ntile = 0; for (int c = 0; c < cs && c < T; ++c) ntile += T - c; }
        else if (head == 3) ntile = Tr * (Tr + 1) / 2;
        if (ntile == 0) return "";
        // few tiles: 128 x 64 tiles (see k_big_syrk); decided on the largest front of the launch
        static const int small_max = getenv("OKKT_SYRK_SMALL_TILES") ? atoi(getenv("OKKT_SYRK_SMALL_TILES")) : 1500;
        const bool narrow = (head == 0 || head == 3) && dbg_syrk == 0 && (int64_t)ntile * g.cnt <= small_max;
        if (narrow) ntile = head == 0 ? T * (T + 1) : Tr * (Tr + 1);
        if (ntile <= 0) return "";
        const dim3 grid((ntile + 7) / 8 * 8, g.cnt);
        const int wofs = par * GS * NB;
        const bool prof = N.profile && (head == 0 || head == 3);   // the dominant kernel: k_big_syrk<0, kSyrkTrail>
        if (prof) {
          // algorithmic flops of this launch: 2 * K * (lower-triangle entries it updates), summed over fronts
          double fl = 0;
          for (int q = 0; q < g.cnt; ++q) {
            const int s = N.sched_host[g.off + q];
            const int kk = N.sn_k[s], ff = N.sn_f[s];
            const int j0 = stepA * NB;
            if (j0 >= kk) continue;
            if (head == 1 && tstep * NB >= kk) continue;
            const double K = std::min(npan * NB, kk - j0);
            const double t0 = std::min(tstep * NB, kk);
            const double remq = ff - t0;
            const double w2 = std::min<double>(cs * 128, remq);          // columns of the look-ahead head
            const double remr = remq - w2;
            if (head == 1) { const double w = std::min<double>(t0 + NB, kk) - t0; fl += 2.0 * K * (w * remq - w * (w - 1.0) / 2.0); }
            else if (head == 2) fl += 2.0 * K * (w2 * remq - w2 * (w2 - 1.0) / 2.0);
            else if (head == 3) fl += K * remr * (remr + 1.0);
            else fl += K * remq * (remq + 1.0);
          }
          if (N.prof_used + 2 > N.prof_events.size())
            for (int q = 0; q < 512; ++q) { hipEvent_t ev; OKKT_HIP_TRY(hipEventCreate(&ev)); N.prof_events.push_back(ev); }
          N.prof_flops.push_back(fl);
          OKKT_HIP_TRY(hipEventRecord(N.prof_events[N.prof_used++], sst));
        }
#define OKKT_SYRK(D, H) hipLaunchKernelGGL((k_big_syrk<D, H>), grid, dim3(kSyrkNW * 64), syrk_lds_bytes(kSyrkStages), sst, P, list, stepA, npan, tstep, NB, wofs, csplit)
        const int csplit = head == 1 ? sub : (head == 0 ? 0 : cs);
        if (head == 1) OKKT_SYRK(0, kSyrkPanel);
        else if (head == 2) OKKT_SYRK(0, kSyrkAhead);
        else if (narrow && dbg_syrk == 96) hipLaunchKernelGGL((k_big_syrk<16, kSyrkTrail, 64>), grid, dim3(kSyrkNW * 64), syrk_lds_bytes(kSyrkStages), sst, P, list, stepA, npan, tstep, NB, wofs, csplit);
        else if (narrow) hipLaunchKernelGGL((k_big_syrk<0, kSyrkTrail, 64>), grid, dim3(kSyrkNW * 64), syrk_lds_bytes(kSyrkStages), sst, P, list, stepA, npan, tstep, NB, wofs, csplit);
        else switch (dbg_syrk) {   // timing-only ablations of the trailing update (OKKT_DEBUG_SYRK): wrong outputs
          case 81: OKKT_SYRK(1, kSyrkTrail); break;    // no C load
          case 82: OKKT_SYRK(2, kSyrkTrail); break;    // no MFMA
          case 85: OKKT_SYRK(5, kSyrkTrail); break;    // no C load, no store
          case 93: OKKT_SYRK(13, kSyrkTrail); break;   // MFMA + LDS reads only
          case 96: OKKT_SYRK(16, kSyrkTrail); break;   // correct results + per-workgroup phase ticks (printed by okkt_get_profile under OKKT_DEBUG_SYRK_LOG)
          case 112: OKKT_SYRK(48, kSyrkTrail); break;  // the same with the C tile waited for before the first operand chunk is requested
          default: OKKT_SYRK(0, kSyrkTrail); break;
        }
#undef OKKT_SYRK
        if (prof) OKKT_HIP_TRY(hipEventRecord(N.prof_events[N.prof_used++], sst));
        return "";
      };
      auto rem_rows = [&](int step) { return g.maxf - step * NB; };
      // rows below the diagonal block of `step`, 64-row blocks [blk_lo, blk_lo + blk_cnt) (blk_cnt < 0: to the end); W goes to
      // slot i of the super-step with parity par
      auto launch_trsm = [&](hipStream_t pst, int step, int par, int i, int blk_lo, int blk_cnt) {
        const int rem = g.maxf - step * NB;  // upper bound on rows below the diagonal block
        int nblk = (rem + 63) / 64 - blk_lo;
        if (blk_cnt >= 0) nblk = std::min(nblk, blk_cnt);
        if (rem <= 0 || nblk <= 0) return;
        const dim3 gr(nblk, g.cnt);
        const int nbl = NB / kIB;
        const size_t lds_trsm = ((size_t)nbl * (nbl + 1) / 2 * kIB * kIB + NB) * sizeof(double);
        const int wc = (par * GS + i) * NB;
        switch (nbl) {
          case 1: hipLaunchKernelGGL(k_big_trsm<1>, gr, dim3(256), lds_trsm, pst, P, list, step, wc, blk_lo); break;
          case 2: hipLaunchKernelGGL(k_big_trsm<2>, gr, dim3(256), lds_trsm, pst, P, list, step, wc, blk_lo); break;
          case 3: hipLaunchKernelGGL(k_big_trsm<3>, gr, dim3(256), lds_trsm, pst, P, list, step, wc, blk_lo); break;
          default: hipLaunchKernelGGL(k_big_trsm<4>, gr, dim3(256), lds_trsm, pst, P, list, step, wc, blk_lo); break;
        }
      };
      // the panels of super-step q (block columns [q * GS, (q + 1) * GS)): diag -> trsm, with the in-group
      // "head" update that brings each later panel of the group up to date first
      auto launch_panels = [&](hipStream_t pst, int stepA, int gs, int par) -> std::string {
        for (int i = 0; i < gs && stepA + i < nsteps; ++i) {
          const int step = stepA + i;
          // in-group update of panel i: only its diagonal tile is on the critical path (k_big_diag needs it); the
          // other tiles run on the auxiliary stream beside k_big_diag and are joined before k_big_trsm
          hipEvent_t ev_rest = nullptr;
          if (i > 0) {
            const bool split = ss.aux != nullptr && N.split_head && rem_rows(step) > split_min_rows;   // the two extra stream hops cost more than they hide on small fronts
            if (split) {
              hipEvent_t ev_t;
              std::string e = next_event(&ev_t);
              if (!e.empty() || !(e = next_event(&ev_rest)).empty()) return e;
              OKKT_HIP_TRY(hipEventRecord(ev_t, pst));                         // trsm of panel i - 1 is behind this
              OKKT_HIP_TRY(hipStreamWaitEvent(ss.aux, ev_t, 0));
              if (!(e = launch_syrk(ss.aux, stepA, i, step, 1, par, 0, 2)).empty()) return e;
              OKKT_HIP_TRY(hipEventRecord(ev_rest, ss.aux));
              if (!(e = launch_syrk(pst, stepA, i, step, 1, par, 0, 1)).empty()) return e;
            } else {
              std::string e = launch_syrk(pst, stepA, i, step, 1, par, 0);
              if (!e.empty()) return e;
            }
          }
          {
            // one launch for the diagonal block and the rows below it (k_diag_trsm_fused) unless something has to happen between
            // the two (the rest of a split in-group update is joined before the trsm)
            const int rem_f = g.maxf - step * NB;
            const int ntr = rem_f > 0 ? (rem_f + 63) / 64 : 0;
            if (N.diag2 && (N.fuse_diag_trsm == 1 || (N.fuse_diag_trsm == 2 && pst == st) || (N.fuse_diag_trsm == 3 && pst == st && g.cnt <= fuse_max_fronts)) && N.chain_flags && dbg_stop == 0 && NB == 128 && ev_rest == nullptr && ntr > 0) {
              const int wc = (par * GS + i) * NB;
              static const int drop_fused = getenv("OKKT_DEBUG_DROP_HANDOFF") ? atoi(getenv("OKKT_DEBUG_DROP_HANDOFF")) : 0;   // bit 3 (tests): the diagonal workgroups raise the flags to an epoch below the one the others wait for
              hipLaunchKernelGGL(k_diag_trsm_fused, dim3((1 + ntr) * g.cnt), dim3(384), std::max(lds_diag2, lds_trsm_max), pst, P, list, step, NB, tol, wc,
                                 N.chain_flags, ++N.chain_epoch, g.cnt, ntr, (drop_fused & 8) ? 1 : 0);
              continue;
            }
          }
          if (N.diag2 && dbg_stop == 0) hipLaunchKernelGGL(k_big_diag2, dim3(g.cnt), dim3(384), lds_diag2, pst, P, list, step, NB, tol);
          else hipLaunchKernelGGL(k_big_diag, dim3(g.cnt), dim3(256), lds_diag, pst, P, list, step, NB, tol, dbg_stop);
          if (ev_rest) OKKT_HIP_TRY(hipStreamWaitEvent(pst, ev_rest, 0));
          launch_trsm(pst, step, par, i, 0, -1);
        }
        return "";
      };
      // Look-ahead: the trailing update of super-step q is split into the tile columns of super-step q + 1
      // (head = 2) and the rest (head = 3).  head2(q) and rest(q) both need Panel(q) and rest(q - 1) and are
      // independent of each other; Panel(q + 1) needs head2(q).  So the panel stream (which finds the CUs the
      // main stream's CU mask leaves free: k_big_diag is a lone, latency-bound workgroup) runs
      // head2(q) -> Panel(q + 1)  while the main stream runs rest(q) back to back with rest(q - 1).
      // Used only while the rest is large enough; afterwards everything runs in order on the main stream.
      // Inverses of the diagonal blocks for the solves (solve.hip).  The NB x NB ones (k_big_invert) run on `st` at the end
      // of the level; the kSolveBlock-column ones of the wide fronts run on the auxiliary stream beside the next levels.
      // A front of at least two such blocks starts early: the blocks that are final when it enters its chain-bound tail
      // (fewer than sb_tail_rows rows left, idle CUs) are inverted during the tail, the rest behind the last panel.
      const SolveLevel& SL = slevels[l];
      static const int inv_main = getenv("OKKT_INV_MAIN") ? atoi(getenv("OKKT_INV_MAIN")) : 0;    // experiment: the block inversions of a dataflow level on the handle's stream behind the launch
      hipStream_t inv_st = (ss.aux && !(inv_main && N.dataflow)) ? ss.aux : st;
      const bool inv_early = ss.aux != nullptr && SL.wide_cnt > 0 && g.maxk >= 2 * kSolveBlock && N.sb_tail_rows >= 0;
      int inv_steps_done = 0, inv_blocks_done = 0;
      const size_t lds_inv = ((size_t)(NB + 2) * NB + 3 * kTld * kIB) * sizeof(double);
      auto inv_range = [&](int steps_final, bool last) -> std::string {
        std::string e2;
        static const int skip_inv = getenv("OKKT_DEBUG_SKIP_INV") ? atoi(getenv("OKKT_DEBUG_SKIP_INV")) : 0;   // experiment: no block inverses (the solves are then wrong): what do they cost the factorisation?
        if (skip_inv) return "";
        if (!last) {
          const int blocks_upto = (int)((int64_t)steps_final * NB / kSolveBlock);
          if (blocks_upto <= inv_blocks_done) return "";
          const int steps_upto = (int)(((int64_t)blocks_upto * kSolveBlock + NB - 1) / NB);
          hipEvent_t evs;
          if (!(e2 = next_event(&evs)).empty()) return e2;
          OKKT_HIP_TRY(hipEventRecord(evs, st));
          OKKT_HIP_TRY(hipStreamWaitEvent(inv_st, evs, 0));
          hipLaunchKernelGGL(k_big_invert, dim3(steps_upto - inv_steps_done, g.cnt), dim3(256), lds_inv, inv_st, P, list, NB, inv_steps_done);
          if (!(e2 = solve_invert_enqueue(N, inv_st, SL, inv_blocks_done, blocks_upto)).empty()) return e2;
          inv_steps_done = steps_upto; inv_blocks_done = blocks_upto;
          inv_on_aux = true; N.inv_stream = inv_st;
          return "";
        }
        // Only the solves read these inverses: the whole batch goes to the auxiliary stream behind the level's last panel (round 3;
        // until then the NB x NB ones ran on the handle's stream: 26 us per level on the critical path, 6 % of an S-C3 factorisation)
        if (nsteps > inv_steps_done || SL.wide_cnt) {
          if (inv_st != st) {
            hipEvent_t evs;
            if (!(e2 = next_event(&evs)).empty()) return e2;
            OKKT_HIP_TRY(hipEventRecord(evs, st));
            OKKT_HIP_TRY(hipStreamWaitEvent(inv_st, evs, 0));
            inv_on_aux = true; N.inv_stream = inv_st;
          }
          if (nsteps > inv_steps_done)
            hipLaunchKernelGGL(k_big_invert, dim3(nsteps - inv_steps_done, g.cnt), dim3(256), lds_inv, inv_st, P, list, NB, inv_steps_done);
          if (SL.wide_cnt && !(e2 = solve_invert_enqueue(N, inv_st, SL, inv_blocks_done, 1 << 30)).empty()) return e2;
        }
        return "";
      };
      // the level's inversions are all enqueued on inv_st: the forward sweep of the next solve waits for THIS event when it reaches this
      // level (single-schedule plans; the top schedule of a partitioned plan keeps the one event of numeric_factor_enqueue)
      auto level_inv_event = [&]() -> std::string {
        if (&levels != &N.levels || !N.levels_top.empty() || inv_st == st) return "";
        if (N.inv_level_events.size() < levels.size()) { N.inv_level_events.resize(levels.size(), nullptr); N.inv_level_pending.resize(levels.size(), 0); }
        if (!N.inv_level_events[l]) OKKT_HIP_TRY(hipEventCreateWithFlags(&N.inv_level_events[l], hipEventDisableTiming));
        OKKT_HIP_TRY(hipEventRecord(N.inv_level_events[l], inv_st));
        N.inv_level_pending[l] = 1;
        return "";
      };
      if (use_df) {
        const bool prof = N.profile;
        if (prof) {
          if (N.prof_used + 2 > N.prof_events.size())
            for (int q = 0; q < 512; ++q) { hipEvent_t ev; OKKT_HIP_TRY(hipEventCreate(&ev)); N.prof_events.push_back(ev); }
          N.prof_flops.push_back(g.df_flops);
          OKKT_HIP_TRY(hipEventRecord(N.prof_events[N.prof_used++], st));
        }
        std::string e;
        e = df_launch(N, P, g, st, tol);
        if (!e.empty()) return e;
        if (prof) OKKT_HIP_TRY(hipEventRecord(N.prof_events[N.prof_used++], st));
        if (!(e = inv_range(nsteps, true)).empty()) return e;
        if (!(e = level_inv_event()).empty()) return e;
        continue;
      }
      int gs_cur = gs_at(0), par = 0;
      std::string e = launch_panels(st, 0, gs_cur, par);
      if (!e.empty()) return e;
      hipEvent_t ev_panel = nullptr;   // set while the panels of the current super-step are on the panel stream
      for (int stepA = 0; stepA < nsteps;) {
        const int stepB = stepA + gs_cur;                          // first block column of the next super-step
        const bool more = stepB < nsteps;
        const int gs_next = more ? gs_at(stepB) : gs_cur;
        const bool la = seg_la && la_at(stepA, gs_cur);
        if (la) {
          hipEvent_t eva, evp;
          if (!(e = next_event(&eva)).empty() || !(e = next_event(&evp)).empty()) return e;
          OKKT_HIP_TRY(hipEventRecord(eva, st));                     // rest(q - 1) (and Panel(0)) are behind this
          OKKT_HIP_TRY(hipStreamWaitEvent(ss.panel, eva, 0));
          if (!(e = launch_syrk(ss.panel, stepA, gs_cur, stepB, 2, par, gs_next)).empty()) return e;
          if (!(e = launch_panels(ss.panel, stepB, gs_next, par ^ 1)).empty()) return e;
          OKKT_HIP_TRY(hipEventRecord(evp, ss.panel));
          if (ev_panel) OKKT_HIP_TRY(hipStreamWaitEvent(st, ev_panel, 0));
          if (!(e = launch_syrk(st, stepA, gs_cur, stepB, 3, par, gs_next)).empty()) return e;
          ev_panel = evp;
        } else {
          if (ev_panel) { OKKT_HIP_TRY(hipStreamWaitEvent(st, ev_panel, 0)); ev_panel = nullptr; }
          // columns below stepB * NB are final behind this point of `st`: once the front is in its chain-bound tail, the
          // inversion of the finished super-blocks starts on the auxiliary stream
          // (every block that has become final since the last call: in the tail the CUs are idle anyway, and the first solve
          // then only waits for the last, partial block)
          if (inv_early && rem_rows(stepB) < N.sb_tail_rows && !(e = inv_range(stepB, false)).empty()) return e;
          if (!(e = launch_syrk(st, stepA, gs_cur, stepB, 0, par, 0)).empty()) return e;
          if (more && !(e = launch_panels(st, stepB, gs_next, par ^ 1)).empty()) return e;
        }
        stepA = stepB; gs_cur = gs_next; par ^= 1;
      }
      // full inverses of the diagonal blocks (for the solves): the remaining block columns of the level in one launch
      if (!(e = inv_range(nsteps, true)).empty()) return e;
      if (!(e = level_inv_event()).empty()) return e;
      if (seg_la) {   // join the handle's stream
        hipEvent_t evj;
        if (!(e = next_event(&evj)).empty()) return e;
        OKKT_HIP_TRY(hipEventRecord(evj, st));
        OKKT_HIP_TRY(hipStreamWaitEvent(ss.main, evj, 0));
      }
    }
  }
  return "";
}


std::string numeric_solve_enqueue(Numeric& N, int R) {
  std::string e;
  static const int dbg_phase = getenv("OKKT_DEBUG_SOLVE_PHASE") ? atoi(getenv("OKKT_DEBUG_SOLVE_PHASE")) : 0;   // 1: forward sweep only (z = D^-1 L^-1 P b)
  if (!(e = solve_fwd_enqueue(N, 0, R)).empty()) return e;
  if (!(e = solve_fwd_enqueue(N, 1, R)).empty()) return e;
  if (dbg_phase == 1) {   // the caller reads the sweep's result where it expects the solution
    OKKT_HIP_TRY(hipMemcpyAsync(N.d.xwork, N.d.zwork, (size_t)N.d.n * kMaxRhs * sizeof(double), hipMemcpyDeviceToDevice, N.stream));
    return "";
  }
  if (!(e = solve_bwd_enqueue(N, 1, R)).empty()) return e;
  return solve_bwd_enqueue(N, 0, R);
}

// ---- multi-GPU exchange helpers (contribution blocks / vectors of the cut, solution pieces) -------------
__global__ void k_pack_cb(DevPlan P, const int* __restrict__ bnd, const int64_t* __restrict__ off, const int* __restrict__ owner,
                          int part, int unpack, double* __restrict__ buf) {
  // one workgroup per boundary front: its r x r contribution block <-> a dense r x r slot of the buffer.
  // Packing writes EVERY slot: the owner its block, everybody else zeros -- the reduce(sum) over the parts is exact and
  // the buffer needs no zero fill beforehand (no ordering hazard between a caller's fill and this stream)
  const int s = bnd[blockIdx.x];
  const bool mine = owner[s] == part;
  if (unpack && mine) return;
  const int k = P.sn_col0[s + 1] - P.sn_col0[s];
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int r = f - k;
  double* F = P.arena + P.front_pos[s] + (size_t)k * f + k + P.cb_shift[s];
  double* B = buf + off[blockIdx.x];
  for (int64_t idx = threadIdx.x; idx < (int64_t)r * r; idx += blockDim.x) {
    const int j = (int)(idx / r), i = (int)(idx - (int64_t)j * r);
    if (unpack) { if (i >= j) F[(size_t)j * f + i] = B[idx]; }
    else B[idx] = (mine && i >= j) ? F[(size_t)j * f + i] : 0.0;
  }
}
__global__ void k_pack_cv(DevPlan P, const int* __restrict__ bnd, const int64_t* __restrict__ off, const int* __restrict__ owner,
                          int part, int unpack, double* __restrict__ buf) {
  const int s = bnd[blockIdx.x];
  const bool mine = owner[s] == part;
  if (unpack && mine) return;
  const int r = (int)(P.rel_ptr[s + 1] - P.rel_ptr[s]);
  double* cv = P.cv + P.cv_pos[s];
  double* B = buf + off[blockIdx.x];
  for (int i = threadIdx.x; i < r; i += blockDim.x) { if (unpack) cv[i] = B[i]; else B[i] = mine ? cv[i] : 0.0; }
}
// mode 0: buf[col] = xwork[col] (all);  1: xwork[col] = buf[col] for top columns only;
// mode 2: sol[perm[col]] = owned(col) ? xwork[col] : 0  (owned: my part, or top when I am part 0)
__global__ void k_exchange_x(int n, const int* __restrict__ col_owner, const int* __restrict__ perm, int part, int mode,
                             double* __restrict__ xwork, double* __restrict__ buf, const int* __restrict__ top_cols, int ntop) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (mode == 0) { if (c < ntop) buf[c] = xwork[top_cols[c]]; return; }      // the separator solution, packed: only the top's columns travel
  if (mode == 1) { if (c < ntop) xwork[top_cols[c]] = buf[c]; return; }
  if (c >= n) return;
  const int o = col_owner[c];
  buf[perm[c]] = (o == part || (o == -1 && part == 0)) ? xwork[c] : 0.0;
}

__global__ void k_sum_counts(const unsigned long long* __restrict__ counters, long long* __restrict__ out) {
  const int c = threadIdx.x;
  if (c == 4) out[4] = (long long)counters[5];      // the time-out word of the in-launch waits: the sharded paths must fail on it, not retry (advisor, round 5)
  if (c >= 4) return;
  unsigned long long sum = 0;
  for (int q = 0; q < kCountSlots; ++q) sum += counters[(size_t)q * kCountStride + c];
  out[c] = (long long)sum;
}
void numeric_sum_counts_device(Numeric& N, long long* d_out4) {
  hipLaunchKernelGGL(k_sum_counts, dim3(1), dim3(64), 0, N.stream, N.d.counters, d_out4);
}

std::string numeric_dist_pack(Numeric& N, int what, int unpack, double* d_buf) {
  if (!N.d.bnd || N.n_boundary == 0) return "";
  if (what == 0) hipLaunchKernelGGL(k_pack_cb, dim3(N.n_boundary), dim3(256), 0, N.stream, N.d, N.d.bnd, N.d.bnd_cb, N.d.sn_owner, N.part_id, unpack, d_buf);
  else hipLaunchKernelGGL(k_pack_cv, dim3(N.n_boundary), dim3(256), 0, N.stream, N.d, N.d.bnd, N.d.bnd_cv, N.d.sn_owner, N.part_id, unpack, d_buf);
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}
std::string numeric_dist_x(Numeric& N, int mode, double* d_buf) {
  const int n = N.d.n;
  const int cnt = mode == 2 ? n : N.n_top_cols;
  if (cnt) hipLaunchKernelGGL(k_exchange_x, dim3((cnt + 255) / 256), dim3(256), 0, N.stream, n, N.d.col_owner, N.d.perm, N.part_id, mode, N.d.xwork, d_buf, N.d.top_cols, N.n_top_cols);
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}

void launch_set_shift(const Numeric& N, double delta, int64_t nshift) {
  const int n = N.d.n;
  if (n) hipLaunchKernelGGL(k_set_shift, dim3((n + 255) / 256), dim3(256), 0, N.stream, n, N.d.perm, delta, (int)nshift, N.d.diagadd);
}

}  // namespace okkt
