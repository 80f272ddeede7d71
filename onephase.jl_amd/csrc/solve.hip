// Triangular solves of the multifrontal LDL^T, gfx950:  x = P' L^-T D^-1 L^-1 P b  for up to kMaxRhs right-hand sides at once
// (`F \ b`, /root/reference/src/linear_system_solvers/julia.jl:99-113; the three refinement solves of schur.jl:158-175).
//
// The sweeps are HBM-bound (2 * 8 * nnz(L) bytes per solve) and, on the big fronts, a chain of dependent steps; what costs
// time is the NUMBER of dependent launches and how much of L each one streams.  Design:
//   * every big front owns explicit inverses of its unit-lower diagonal blocks: fronts with k <= NB pivot columns use the
//     NB x NB inverse that the factorisation leaves in `invl`; wider fronts get inverses of their kSB-column diagonal blocks
//     (k_xinv_*: recursive doubling X = [[A^-1, 0], [-B^-1 C A^-1, B^-1]] from the NB x NB inverses, FP64 MFMA products),
//     computed on the auxiliary stream beside the factorisation.  A sweep over a front is then one product with X and one
//     tall GEMV per kSB columns -- two launches -- instead of two per 128 columns;
//   * a level's fronts with k <= NB take ONE forward launch: every workgroup rebuilds y = X w redundantly (128 x 128),
//     assembles its own rows from the children's contribution vectors and applies its 64 rows of the panel;
//   * all kernels carry R right-hand sides through one pass over L (R = 1, 2, 4): the panel values are loaded once and
//     multiplied with R vectors held in LDS -- a batched okkt_solve streams the factor once per batch, not once per rhs;
//   * panel reads are 16 bytes per lane (two rows), 8 - 16 loads in flight per lane.
// Results are deterministic: every output entry is owned by one workgroup that sums in a fixed order.
#include "numeric.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace okkt {

#define OKKT_HIP_TRY(expr)                                                         \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess)                                                         \
      return std::string(#expr) + ": " + hipGetErrorString(e__);                   \
  } while (0)

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

constexpr int kSB = kSolveBlock;        // columns of an explicitly inverted diagonal block
#ifndef OKKT_SOLVE_SPLITS
#define OKKT_SOLVE_SPLITS 8
#endif
constexpr int kCS = OKKT_SOLVE_SPLITS;  // splits of a block product (partial vectors summed by the consumer in a fixed order)
static_assert(kSB / kCS >= 128 && (kCS & (kCS - 1)) == 0, "the block-product kernels need column splits of at least 128 columns (64 gave wrong results: found in round 3)");
// sum of the kCS partial values p[0], p[kSB], ... (pairwise, fixed order)
__device__ __forceinline__ double sum_splits(const double* p) {
  double t[kCS];
#pragma unroll
  for (int q = 0; q < kCS; ++q) t[q] = p[(size_t)q * kSB];
#pragma unroll
  for (int w = 1; w < kCS; w *= 2)
#pragma unroll
    for (int q = 0; q + w < kCS; q += 2 * w) t[q] += t[q + w];
  return t[0];
}

__device__ __forceinline__ double sum_splits_agent(const double* p) {
  double t[kCS];
#pragma unroll
  for (int q = 0; q < kCS; ++q) t[q] = __hip_atomic_load(p + (size_t)q * kSB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int w = 1; w < kCS; w *= 2)
#pragma unroll
    for (int q = 0; q + w < kCS; q += 2 * w) t[q] += t[q + w];
  return t[0];
}

__device__ __forceinline__ int round128(int v) { return (v + 127) & ~127; }
// block b of a front with k pivot columns: first column, width, leading dimension and storage offset of its inverse
__device__ __forceinline__ void xblock(int k, int b, int& c0, int& kb, int& ld, int64_t& off) {
  c0 = b * kSB;
  kb = min(kSB, k - c0);
  ld = kb >= kSB ? kSB : round128(kb);
  off = (int64_t)b * kSB * kSB;
}

// forward right-hand-side entry of front row r (before any block of this front has been applied): the permuted rhs on the
// pivot rows plus the children's contribution vectors through the inverted extend-add lists (fixed order)
template <int R>
__device__ __forceinline__ void fwd_gather(const DevPlan& P, int64_t gcb, int col0, int k, int r, double (&w)[R]) {
#pragma unroll
  for (int q = 0; q < R; ++q) w[q] = r < k ? P.xwork[(size_t)q * P.xw_stride + col0 + r] : 0.0;
  // the items are added strictly in list order, but eight of them have their position and value loads in flight together
  // (one precomputed position per item: two dependent round trips per batch instead of four per item)
  const int64_t q0 = P.ea_ptr[gcb + r], q1 = P.ea_ptr[gcb + r + 1];
  for (int64_t e = q0; e < q1; e += 8) {
    int64_t pos[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) pos[u] = P.ea_pos[min(e + u, q1 - 1)];
    double v[8][R];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < R; ++q) v[u][q] = P.cv[(size_t)q * P.cv_stride + pos[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e + u < q1) {
#pragma unroll
        for (int q = 0; q < R; ++q) w[q] += v[u][q];
      }
  }
}
// where the running forward vector of front row r lives: pivot rows in xwork, the rows below in the front's contribution vector
template <int R>
__device__ __forceinline__ double* fwd_slot(const DevPlan& P, int s, int col0, int k, int r, int q) {
  return r < k ? P.xwork + (size_t)q * P.xw_stride + col0 + r : P.cv + (size_t)q * P.cv_stride + P.cv_pos[s] + (r - k);
}

// ------------------------------------------------------------------------------------------------------------------
// explicit inverses of the kSB-column diagonal blocks
// ------------------------------------------------------------------------------------------------------------------
// The NB x NB diagonal tiles of X_b <- the NB x NB inverses of the factorisation (one workgroup per tile).  Everything else
// of the lower triangle is overwritten by the doubling levels below, the strict upper tiles and the padding rows / columns
// beyond the block are zero from the allocation and are never written.
__global__ __launch_bounds__(256) void k_xinv_init(DevPlan P, const int* __restrict__ list, int b0, int nbk, int NB) {
  const int s = list[blockIdx.z / nbk];
  const int b = b0 + (int)(blockIdx.z % nbk);
  const int k = P.sn_col0[s + 1] - P.sn_col0[s];
  int c0, kb, ld; int64_t off;
  xblock(k, b, c0, kb, ld, off);
  if (kb <= 0 || (P.solve_mid && k <= P.solve_mid)) return;      // (nobody reads the explicit inverse of a front that goes through block substitution)
  const int ti = blockIdx.x;
  if (ti * NB >= kb) return;
  double* X = P.xinv + P.xinv_pos[s] + off;
  const double* Inb = P.invl + P.invl_pos[s] + (size_t)(c0 / NB + ti) * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    const int i = e % NB, j = e / NB;
    X[(size_t)(ti * NB + j) * ld + ti * NB + i] = Inb[i + (size_t)j * NB];
  }
}

// One doubling level s -> 2s of the recursive inverse.  Pair p of block b: A = X[2ps.., 2ps..] (s x s), B = X[(2p+1)s.., (2p+1)s..]
// (M x M, M <= s), C = L[(2p+1)s.., 2ps..] of the front.  PHASE 0: T = C * A into xtmp; PHASE 1: X21 = -B * T.
// 64 x 64 tile per workgroup, 32 x 32 per wave as 2 x 2 v_mfma_f64_16x16x4; operands through LDS in k-chunks of 16.
// The triangular operand bounds the k-range of a tile (A lower: k >= n0; B lower: k < m0 + 64).
template <int PHASE>
__global__ __launch_bounds__(256) void k_xinv_gemm(DevPlan P, const int* __restrict__ list, int b0, int nbk, int sz) {
  __shared__ double As[16][68], Bs[16][68];
  const int s = list[blockIdx.z / nbk];
  const int b = b0 + (int)(blockIdx.z % nbk);
  const int k = P.sn_col0[s + 1] - P.sn_col0[s];
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  int c0, kb, ld; int64_t off;
  xblock(k, b, c0, kb, ld, off);
  if (kb <= 0 || (P.solve_mid && k <= P.solve_mid)) return;
  const int p = blockIdx.y;
  const int r0 = (2 * p + 1) * sz, q0 = 2 * p * sz;       // rows of the lower half, columns of the left half (inside the block)
  if (r0 >= kb) return;
  const int M = min(sz, kb - r0), N = sz;
  const int tn = (N + 63) / 64;
  const int tile_m = (int)blockIdx.x / tn, tile_n = (int)blockIdx.x % tn;
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  if (m0 >= M) return;
  double* X = P.xinv + P.xinv_pos[s] + off;
  double* T = P.xtmp + P.xinv_pos[s] + off;
  const double* Am; int lda;
  const double* Bm; int ldb;
  double* Cm; int ldc;
  int kbeg, kend;
  if (PHASE == 0) {
    Am = P.arena + P.front_pos[s] + (size_t)(c0 + q0) * f + (c0 + r0); lda = f;     // C block of L
    Bm = X + (size_t)q0 * ld + q0; ldb = ld;                                        // A^-1, lower triangular
    Cm = T + (size_t)q0 * ld + r0; ldc = ld;
    kbeg = n0; kend = sz;
  } else {
    Am = X + (size_t)r0 * ld + r0; lda = ld;                                        // B^-1, lower triangular
    Bm = T + (size_t)q0 * ld + r0; ldb = ld;
    Cm = X + (size_t)q0 * ld + r0; ldc = ld;
    kbeg = 0; kend = min(M, m0 + 64);
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int wm0 = (wv & 1) * 32, wn0 = (wv >> 1) * 32;
  d4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
  for (int k0 = kbeg & ~15; k0 < kend; k0 += 16) {
    // stage: A tile 64 rows x 16 k (columns of A are contiguous), B tile 16 k x 64 columns
    {
      const int m = tid & 63;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = (tid >> 6) + 4 * j;
        const bool ok = m0 + m < M && k0 + kk >= kbeg && k0 + kk < kend;
        const double v = Am[(size_t)min(k0 + kk, kend - 1) * lda + min(m0 + m, M - 1)];
        As[kk][m] = ok ? v : 0.0;
      }
      const int kk = tid & 15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = (tid >> 4) + 16 * j;
        const bool ok = k0 + kk >= kbeg && k0 + kk < kend && n0 + n < N;
        const double v = Bm[(size_t)min(n0 + n, N - 1) * ldb + min(k0 + kk, kend - 1)];
        Bs[kk][n] = ok ? v : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = As[4 * q + l4][wm0 + 16 * i + l15];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = Bs[4 * q + l4][wn0 + 16 * j + l15];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // accumulator register `reg` of tile (i, j): row 16 i + l4 + 4 reg, column 16 j + l15
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + wm0 + 16 * i + l4 + 4 * reg, n = n0 + wn0 + 16 * j + l15;
        if (m < M && n < N) Cm[(size_t)n * ldc + m] = PHASE == 0 ? acc[i][j][reg] : -acc[i][j][reg];
      }
}

// ---- in-launch hand-offs between the workgroups of one front (round 3): a level's two dependent launches become one.
// Producer workgroups have LOWER block indices than their consumers (dispatch order = dependency order, so a waiting
// consumer never keeps its producer from being scheduled); the protocol is the MI355X guide's: every storing wave drains its
// stores, the workgroup meets at a barrier, one lane releases at agent scope and then raises the flag / counter; the consumer
// polls with relaxed agent-scope loads (bounded: a hand-off that never arrives ends the wait instead of hanging the GPU and
// raises the plan's time-out word, which turns the solution into NaN in k_permute_out_r), acquires once, and only then reads the data.
// The handed-off bytes are a few hundred doubles per front: they are written and read with agent-scope (sc1) accesses, which
// bypass the CU's L1 and are served by the memory side -- no release / acquire fences (an agent release writes back the whole L2,
// an acquire invalidates the whole L1: 2 - 7 us each, per workgroup, measured as a net loss on the fused sweeps).
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void front_signal_store(int* flag, int value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its sc1 stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void front_signal_add(int* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Arrival counters shared by many producers and many consumers of one launch: value = (epoch << 20) + arrivals.  Every producer
// first raises the counter to its launch's epoch base (atomic max: idempotent, whoever comes first sets it), then adds one;
// the consumers of that launch wait for base + expected.  Epochs grow with every fused launch of the handle, so nothing is reset.
__device__ __forceinline__ void front_arrive64(unsigned long long* counter, unsigned long long epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_max(counter, epoch << 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// `tmo` (slot 0, word 5 of the plan's counters): raised when a wait runs into its bound; the last kernel of a solve
// (k_permute_out_r) then returns NaN instead of a solution that was computed from data that had not arrived
__device__ __forceinline__ void front_wait64(const unsigned long long* counter, unsigned long long epoch, int expected, unsigned long long* tmo = nullptr) {
  if (threadIdx.x == 0) {
    const unsigned long long want = (epoch << 20) + (unsigned long long)expected;
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
    if (spins >= (1 << 22) && tmo) atomicExch(tmo, 1ull);
  }
  __syncthreads();
}
// waits until *flag >= value (monotonic flags); the payload is then read with ld_agent only
__device__ __forceinline__ void front_wait(const int* flag, int value, unsigned long long* tmo = nullptr) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
    if (spins >= (1 << 22) && tmo) atomicExch(tmo, 1ull);
  }
  __syncthreads();
}

// Flow sweeps of the wide fronts (k_fwd_wide_flow / k_bwd_wide_flow): wave 0 waits until the tile words ver[t0 .. t0 + nt) have
// reached `want` (nt <= 62) and, with a counter, until the arrival counter has reached `cwant` -- one condition per lane, one
// ballot per poll; bounded like the other waits.  The payload travels with agent-scope accesses only (no fences, see above).
__device__ __forceinline__ void flow_wait(const unsigned long long* ver, int t0, int nt, unsigned long long want, const unsigned long long* counter, unsigned long long cwant,
                                          unsigned long long* tmo) {
  if (threadIdx.x < 64) {
    const int ln = threadIdx.x;
    const unsigned long long* addr = ln < nt ? ver + t0 + ln : ((ln == 63 && counter) ? counter : nullptr);
    const unsigned long long need = ln < nt ? want : cwant;
    int spins = 0;
    for (;;) {
      const bool ok = addr == nullptr || __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
      if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
      if (++spins >= (1 << 22)) {
        if (ln == 0 && tmo) atomicExch(tmo, 1ull);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}
// ... and publishes tile words once every wave's agent-scope stores have been acknowledged
__device__ __forceinline__ void flow_publish(unsigned long long* ver, int t0, int nt, unsigned long long value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < nt) __hip_atomic_store(ver + t0 + threadIdx.x, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------------------------
// small fronts (one workgroup per task of fronts, the front's vector in LDS), R right-hand sides
// ------------------------------------------------------------------------------------------------------------------
// FLOW: the tasks of several levels in one launch (see k_front_small<.., FLOW> in numeric.hip): a task waits for the flags of its
// children tasks, contribution vectors travel with agent-scope accesses, flags[root] = epoch when the task is done.
template <int TPB, int R, bool FLOW = false>
__global__ __launch_bounds__(TPB) void k_fs_small(DevPlan P, const int* __restrict__ list, int ldw, int* __restrict__ flags = nullptr, int epoch = 0, int wait_epoch = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];      // [R][ldw]
  const int tid = threadIdx.x;
  const int s_root = list[blockIdx.x];
  const int t_lo = P.task_lo[s_root];
  for (int s = t_lo; s <= s_root; ++s) {      // the task's fronts, children first
    const int col0 = P.sn_col0[s];
    const int k = P.sn_col0[s + 1] - col0;
    const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
    const double* L = P.arena + P.front_pos[s];
    for (int i = tid; i < f; i += TPB)
#pragma unroll
      for (int q = 0; q < R; ++q) sm[q * ldw + i] = i < k ? P.xwork[(size_t)q * P.xw_stride + col0 + i] : 0.0;
    __syncthreads();
    for (int64_t c = P.child_ptr[s]; c < P.child_ptr[s + 1]; ++c) {
      const int ch = P.children[c];
      const int rc = (int)(P.rel_ptr[ch + 1] - P.rel_ptr[ch]);
      const int* rl = P.rel + P.rel_ptr[ch];
      const double* cvc = P.cv + P.cv_pos[ch];
      if (FLOW && ch < t_lo) front_wait(flags + ch, wait_epoch, P.counters + 5);
      for (int ii = tid; ii < rc; ii += TPB) {
        const int d = rl[ii];
#pragma unroll
        for (int q = 0; q < R; ++q) sm[q * ldw + d] += FLOW ? ld_agent(cvc + (size_t)q * P.cv_stride + ii) : cvc[(size_t)q * P.cv_stride + ii];
      }
      __syncthreads();
    }
    // unit lower triangular k x k
    for (int j = 0; j < k; ++j) {
      const double* col = L + (size_t)j * f;
      for (int i = j + 1 + tid; i < k; i += TPB) {
        const double l = col[i];
#pragma unroll
        for (int q = 0; q < R; ++q) sm[q * ldw + i] -= l * sm[q * ldw + j];
      }
      __syncthreads();
    }
    // rows below the pivot block: contribution vector for the parent
    double* cvs = P.cv + P.cv_pos[s];
    for (int i = k + tid; i < f; i += TPB) {
      double acc[R];
#pragma unroll
      for (int q = 0; q < R; ++q) acc[q] = sm[q * ldw + i];
      for (int j = 0; j < k; ++j) {
        const double l = L[(size_t)j * f + i];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] -= l * sm[q * ldw + j];
      }
#pragma unroll
      for (int q = 0; q < R; ++q) { if (FLOW) st_agent(cvs + (size_t)q * P.cv_stride + i - k, acc[q]); else cvs[(size_t)q * P.cv_stride + i - k] = acc[q]; }
    }
    // z = D^-1 y
    for (int j = tid; j < k; j += TPB) {
      const double d = P.dvals[col0 + j];
#pragma unroll
      for (int q = 0; q < R; ++q) P.zwork[(size_t)q * P.xw_stride + col0 + j] = sm[q * ldw + j] / d;
    }
    if (FLOW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    __syncthreads();
  }
  if (FLOW) front_signal_store(flags + s_root, epoch);
}

// FLOW, backward: block b takes list[gridDim.x - 1 - b] (parents before children in dispatch order), waits for the task that
// holds its root's parent front and reads the ancestors' solution with agent-scope loads.
template <int TPB, int R, bool FLOW = false>
__global__ __launch_bounds__(TPB) void k_bs_small(DevPlan P, const int* __restrict__ list, int ldw, int* __restrict__ flags = nullptr, int epoch = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];      // [R][ldw]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = TPB / 64;
  const int s_root = list[FLOW ? gridDim.x - 1 - blockIdx.x : blockIdx.x];
  if (FLOW) {
    const int up = P.unit_parent[s_root];
    if (up >= 0) front_wait(flags + up, epoch, P.counters + 5);
  }
  for (int s = s_root; s >= P.task_lo[s_root]; --s) {      // the task's fronts, parents first
    const int col0 = P.sn_col0[s];
    const int k = P.sn_col0[s + 1] - col0;
    const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
    const double* L = P.arena + P.front_pos[s];
    const int* rows = P.rows + P.row_ptr[s];
    for (int i = tid; i < f; i += TPB) {
      const int g = rows[i];     // rows[i] = col0 + i for i < k: this front's z; beyond: the ancestors' solution
      const double* src = i < k ? P.zwork : P.xwork;
#pragma unroll
      for (int q = 0; q < R; ++q) sm[q * ldw + i] = (FLOW && i >= k) ? ld_agent(src + (size_t)q * P.xw_stride + g) : src[(size_t)q * P.xw_stride + g];
    }
    __syncthreads();
    // rhs_j = z_j - sum_{i >= k} L[i, j] * x_i
    for (int j = wv; j < k; j += NW) {
      const double* col = L + (size_t)j * f;
      double acc[R];
#pragma unroll
      for (int q = 0; q < R; ++q) acc[q] = 0.0;
      for (int i = k + lane; i < f; i += 64) {
        const double l = col[i];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] += l * sm[q * ldw + i];
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        double a = acc[q];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0) sm[q * ldw + j] -= a;
      }
    }
    __syncthreads();
    // unit upper triangular (L11^T) k x k, column oriented
    for (int j = k - 1; j >= 0; --j) {
      for (int i = tid; i < j; i += TPB) {
        const double l = L[(size_t)i * f + j];
#pragma unroll
        for (int q = 0; q < R; ++q) sm[q * ldw + i] -= l * sm[q * ldw + j];
      }
      __syncthreads();
    }
    for (int j = tid; j < k; j += TPB)
#pragma unroll
      for (int q = 0; q < R; ++q) { if (FLOW) st_agent(P.xwork + (size_t)q * P.xw_stride + col0 + j, sm[q * ldw + j]); else P.xwork[(size_t)q * P.xw_stride + col0 + j] = sm[q * ldw + j]; }
    if (FLOW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    __syncthreads();
  }
  if (FLOW) front_signal_store(flags + s_root, epoch);
}


// ------------------------------------------------------------------------------------------------------------------
// big fronts, forward
// ------------------------------------------------------------------------------------------------------------------
// Fronts with k <= NB <= 128 pivot columns, two launches per level.  (1) one workgroup per front: assemble w_K, y = X w_K with
// the NB x NB inverse of the factorisation, store z = y / d and y.  (2) the rows below the pivot block, 128 per workgroup
// (two per lane, 16-byte loads; wave g takes the columns g, g + 4, ...): w[r] = assembled rhs - sum_c L[r, c] y[c] -> the
// front's contribution vector.  (One fused launch with y recomputed by every workgroup was 2 x slower: 128 KiB of X and the
// w_K gathers per 64 rows.)
template <int R, bool SC1>
__device__ __forceinline__ void fwd_thin_y_body(const DevPlan& P, int s, int NB, double (*wk)[128]) {
  const int tid = threadIdx.x;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int64_t gcb = P.bigcol_base[s];
  const double* X = P.invl + P.invl_pos[s];
  // X first: its loads do not depend on the gather
  const int c = tid >> 1, h = tid & 1;
  const int cc = min(c, NB - 1);
  double v[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) v[q] = X[cc + (size_t)min(h * 64 + q, NB - 1) * NB];
  if (tid < 128) {
    double w[R];
    if (tid < k) fwd_gather<R>(P, gcb, col0, k, tid, w);
#pragma unroll
    for (int q = 0; q < R; ++q) wk[q][tid] = tid < k ? w[q] : 0.0;
  }
  __syncthreads();
  double* yt = P.ythin + P.ythin_pos[s];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) a += (h * 64 + q < NB) ? v[q] * wk[r][h * 64 + q] : 0.0;
    a += __shfl_xor(a, 1, 64);
    if (h == 0 && c < NB) {
      if (SC1) st_agent(&yt[r * 128 + c], c < k ? a : 0.0); else yt[r * 128 + c] = c < k ? a : 0.0;
      if (c < k) P.zwork[(size_t)r * P.xw_stride + col0 + c] = a / P.dvals[col0 + c];
    }
  }
}
template <int R>
__global__ __launch_bounds__(256) void k_fwd_thin_y(DevPlan P, const int* __restrict__ list, int NB) {
  __shared__ double wk[R][128];
  fwd_thin_y_body<R, false>(P, list[blockIdx.x], NB, wk);
}

// rows below the pivot block.  xoff = 0 / flag = NULL: the second launch of the two-launch form; xoff = 1: workgroup blockIdx.x - 1
// of the fused launch, which waits for the front's y flag behind its own panel loads and gathers
template <int R>
__device__ __forceinline__ void fwd_thin_upd_body(const DevPlan& P, int s, int xblk, const int* flag, int epoch, double (*yk)[128], double (*part)[R][128]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int rb = k + xblk * 128;
  if (rb >= f) return;
  const double* F = P.arena + P.front_pos[s];
  const int r0 = rb + 2 * lane;
  const bool pair_ok = f >= 2;
  const int pr = pair_ok ? max(min(r0, f - 2), 0) : 0;
  // the panel entries: wave wv takes the columns wv, wv + 4, ... (k <= 128: at most 32 of them), all in flight
  d2_t lv[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int cq = min(wv + 4 * u, k - 1);
    if (pair_ok) __builtin_memcpy(&lv[u], F + (size_t)cq * f + pr, 16);
    else { lv[u][0] = F[(size_t)cq * f + pr]; lv[u][1] = 0.0; }
  }
  // this thread's row (threads 0..127: one row each), assembled while the panel loads are in flight
  const int64_t gcb = P.bigcol_base[s];
  const int myrow = rb + tid;
  double wr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) wr[r] = 0.0;
  if (tid < 128 && myrow < f) fwd_gather<R>(P, gcb, col0, k, myrow, wr);
  if (flag) front_wait(flag, epoch, P.counters + 5);          // y of this front (fused launch): the loads above are already in flight
  const double* yt = P.ythin + P.ythin_pos[s];
  if (tid < 128)
#pragma unroll
    for (int r = 0; r < R; ++r) yk[r][tid] = flag ? ld_agent(&yt[r * 128 + tid]) : yt[r * 128 + tid];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int cq = wv + 4 * u;
      if (cq < k) { const double y = yk[r][cq]; a0 += lv[u][0] * y; a1 += lv[u][1] * y; }
    }
    part[wv][r][2 * lane] = a0;
    part[wv][r][2 * lane + 1] = a1;
  }
  __syncthreads();
  if (tid < 128 && myrow < f) {
    // pair value index of my row: lane ln = (myrow - rb) / 2 holds rows (pr_ln, pr_ln + 1)
    const int ln = tid >> 1;
    const int r0l = rb + 2 * ln;
    const int prl = pair_ok ? max(min(r0l, f - 2), 0) : 0;
    const int e = myrow - prl;                   // 0 or 1 (the clamped last pair: row f - 1 is value 1)
    if (e >= 0 && e <= 1) {
      double* cvs = P.cv + P.cv_pos[s] + (myrow - k);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int idx = 2 * ln + e;
        cvs[(size_t)r * P.cv_stride] = wr[r] - ((part[0][r][idx] + part[1][r][idx]) + (part[2][r][idx] + part[3][r][idx]));
      }
    }
  }
}

template <int R>
__global__ __launch_bounds__(256) void k_fwd_thin_upd(DevPlan P, const int* __restrict__ list) {
  __shared__ double yk[R][128], part[4][R][128];
  fwd_thin_upd_body<R>(P, list[blockIdx.y], (int)blockIdx.x, nullptr, 0, yk, part);
}
// one launch per level: workgroup 0 of a front computes y and raises the front's flag, the others apply the panel rows
template <int R>
__global__ __launch_bounds__(256) void k_fwd_thin_fused(DevPlan P, const int* __restrict__ list, int NB, int* __restrict__ flags, int epoch) {
  __shared__ double yk[R][128], part[4][R][128];
  const int s = list[blockIdx.y];
  if (blockIdx.x == 0) {
    fwd_thin_y_body<R, true>(P, s, NB, yk);
    front_signal_store(flags + s, epoch);
  } else {
    fwd_thin_upd_body<R>(P, s, (int)blockIdx.x - 1, flags + s, epoch, yk, part);
  }
}

// Wide fronts, block b: partial products of y = X_b w_b.  Workgroup (i, front, cq): rows [64 i, +64) of the block, columns
// [cq * kSB / kCS, +kSB / kCS); X_b is lower triangular, so a workgroup stops at its last row.  The kCS partial vectors are
// summed by the consumer (k_fwd_upd).  In block 0 the right-hand side is assembled on the fly (children's contributions).
// returns false when the workgroup has no row of the block (it then takes no part in the hand-off of the fused launch)
// ver != NULL (flow launch): the right-hand side rows of block b > 0 were written by workgroups of the same launch -- their 32-row
// tiles are awaited (vwant = blocks applied to them) and read with agent-scope loads
template <int R, bool AG>
__device__ __forceinline__ bool fwd_y_body(const DevPlan& P, int s, int b, int bx, int cq, const unsigned long long* ver = nullptr, unsigned long long vwant = 0) {
  constexpr int CW = kSB / kCS;
  __shared__ double wj[R][CW], part[4][R][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  int c0, kb, ld; int64_t off;
  xblock(k, b, c0, kb, ld, off);
  if (kb <= 0) return false;
  const int i0 = bx * 64;
  if (i0 >= kb) return false;
  const int i = i0 + lane;
  double* yp = P.ypart + P.ypart_pos[s] + (ver ? (size_t)b * kMaxRhs * kCS * kSB : 0) + (size_t)cq * kSB;          // [R][kCS][kSB]; a flow launch: one such buffer per block (block b + 1's products run beside block b's far rows)
  const int pend = min(kb, i0 + 64);              // no row of this workgroup reaches beyond its last row
  const int q0 = cq * CW;
  if (q0 >= pend) {
    if (wv == 0 && i < kb)
#pragma unroll
      for (int r = 0; r < R; ++r) { if (AG) st_agent(&yp[(size_t)r * kCS * kSB + i], 0.0); else yp[(size_t)r * kCS * kSB + i] = 0.0; }
    return true;
  }
  const int kq = min(CW, pend - q0);              // columns of this split that matter
  const double* X = P.xinv + P.xinv_pos[s] + off + (size_t)q0 * ld;
  const int ic = min(i, kb - 1);
  const int64_t gcb = P.bigcol_base[s];
  if (ver && b > 0) flow_wait(ver, (c0 + q0) >> 5, ((c0 + q0 + kq - 1) >> 5) - ((c0 + q0) >> 5) + 1, vwant, nullptr, 0ull, P.counters + 5);
  for (int c = tid; c < CW; c += 256) {
    double w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) w[r] = 0.0;
    if (c < kq) {
      if (b == 0) fwd_gather<R>(P, gcb, col0, k, q0 + c, w);
      else {
#pragma unroll
        for (int r = 0; r < R; ++r) w[r] = ver ? ld_agent(&P.xwork[(size_t)r * P.xw_stride + col0 + c0 + q0 + c]) : P.xwork[(size_t)r * P.xw_stride + col0 + c0 + q0 + c];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) wj[r][c] = w[r];
  }
  __syncthreads();
  double a[R];
#pragma unroll
  for (int r = 0; r < R; ++r) a[r] = 0.0;
  for (int p0 = wv * (CW / 4); p0 < min(kq, (wv + 1) * (CW / 4)); p0 += 32) {
    double v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = X[(size_t)min(p0 + q, kq - 1) * ld + ic];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const double vv = p0 + q < kq ? v[q] : 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] += vv * wj[r][min(p0 + q, CW - 1)];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) part[wv][r][lane] = a[r];
  __syncthreads();
  if (wv == 0 && i < kb)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double v = (part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane]);
      if (AG) st_agent(&yp[(size_t)r * kCS * kSB + i], v); else yp[(size_t)r * kCS * kSB + i] = v;
    }
  return true;
}
template <int R>
__global__ __launch_bounds__(256) void k_fwd_y(DevPlan P, const int* __restrict__ list, int b) {
  if (P.solve_mid) { const int s = list[blockIdx.y]; if (P.sn_col0[s + 1] - P.sn_col0[s] <= P.solve_mid) return; }      // k_fwd_mid's fronts
  (void)fwd_y_body<R, false>(P, list[blockIdx.y], b, (int)blockIdx.x, (int)blockIdx.z);
}

// Wide fronts, block b: y = sum of the partials (z = y / d stored by the first workgroup), then the rows below the block:
// w[r] -= sum_p L[r][c0 + p] y[p].  64 rows per workgroup: a half-wave owns 32 row pairs (16-byte loads: two rows per lane),
// the eight half-waves take the columns h, h + 8, ...; sixteen columns are in flight per lane (64 KiB per workgroup), the
// partial sums meet in LDS.  Block 0 assembles the rows it touches on the fly.
constexpr int kUpdRows = 64;
// counter != NULL: a workgroup of the fused launch -- it waits for the partial products of its front before it sums them
// ROWS = rows per workgroup: 64 (two per lane, eight groups of 32 lanes share the columns) or 32 (sixteen groups of 16 lanes) for
// launches that would otherwise leave most CUs idle (the later blocks of a front: rows / 64 workgroups)
// ver != NULL: a workgroup of the flow launch (all blocks of a level's wide fronts in one launch): its rows start at the absolute row
// rb_abs (a multiple of 32), `first` marks the workgroup that stores z, the arrival counter is the block's own (yexp arrivals), the
// 32-row tiles it owns are awaited (vbase + b: the updates of the blocks before b) before the running vector is read, everything
// another workgroup of the launch reads or wrote travels with agent-scope accesses, and the tiles are published (vbase + b + 1)
template <int R, int ROWS = kUpdRows>
__device__ __forceinline__ void fwd_upd_body(const DevPlan& P, int s, int b, int bx, double* sm, const unsigned long long* counter, unsigned long long epoch,
                                             unsigned long long* ver = nullptr, unsigned long long vbase = 0, int rb_abs = -1, int first = -1, long long yexp = -1) {
  constexpr int GW = ROWS / 2, NG = 256 / GW;      // lanes per group, groups
  double* yj = sm;                                 // y[R][kSB], then part[NG][R][ROWS]
  double* part = sm + (size_t)R * kSB;
  const int tid = threadIdx.x, l32 = tid % GW, hw = tid / GW;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  int c0, kb, ld; int64_t off;
  xblock(k, b, c0, kb, ld, off);
  if (kb <= 0) return;
  const int rbeg = c0 + kb;                       // first row below the block
  const int rb = rb_abs >= 0 ? rb_abs : rbeg + bx * ROWS;
  const bool is_first = first >= 0 ? first != 0 : bx == 0;
  if (!is_first && rb >= f) return;
  const int vt0 = rb >> 5, vnt = rb < f ? min(ROWS / 32, (f - rb + 31) >> 5) : 0;
  if (ver) flow_wait(ver, vt0, b > 0 ? vnt : 0, vbase + (unsigned long long)b, counter, (epoch << 20) + (unsigned long long)yexp, P.counters + 5);
  else if (counter) front_wait64(counter, epoch, ((kb + 63) / 64) * kCS, P.counters + 5);
  const double* yp = P.ypart + P.ypart_pos[s] + (ver ? (size_t)b * kMaxRhs * kCS * kSB : 0);
  for (int p = tid; p < kSB; p += 256) {
    const int pc = min(p, kb - 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double y = counter ? sum_splits_agent(yp + (size_t)r * kCS * kSB + pc) : sum_splits(yp + (size_t)r * kCS * kSB + pc);
      yj[r * kSB + p] = p < kb ? y : 0.0;
      if (is_first && p < kb) P.zwork[(size_t)r * P.xw_stride + col0 + c0 + p] = y / P.dvals[col0 + c0 + p];
    }
  }
  __syncthreads();
  if (rb >= f) return;
  // rows (pr, pr + 1): the pair start is clamped so that both loads stay inside the column; the final pass maps them back
  const int r0 = rb + 2 * l32;
  const bool pair_ok = f >= 2;
  const int pr = pair_ok ? max(min(r0, f - 2), 0) : 0;
  const double* Lp = P.arena + P.front_pos[s] + (size_t)c0 * f + pr;
  double a0[R], a1[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { a0[r] = 0.0; a1[r] = 0.0; }
  for (int p0 = hw; p0 < kb; p0 += 16 * NG) {          // columns p0, p0 + NG, ..., p0 + 15 NG
    d2_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int p = min(p0 + NG * u, kb - 1);
      if (pair_ok) __builtin_memcpy(&v[u], Lp + (size_t)p * f, 16);
      else { v[u][0] = Lp[(size_t)p * f]; v[u][1] = 0.0; }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int p = p0 + NG * u;
      if (p < kb) {
#pragma unroll
        for (int r = 0; r < R; ++r) { const double y = yj[r * kSB + p]; a0[r] += v[u][0] * y; a1[r] += v[u][1] * y; }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    part[((size_t)hw * R + r) * ROWS + 2 * l32] = a0[r];
    part[((size_t)hw * R + r) * ROWS + 2 * l32 + 1] = a1[r];
  }
  __syncthreads();
  if (tid < ROWS) {
    const int ln = tid >> 1, e = tid & 1;         // value e of lane ln's pair
    const int r0l = rb + 2 * ln;
    const int prl = pair_ok ? max(min(r0l, f - 2), 0) : 0;
    const int sh = r0l - prl;
    // which front row does pair value e belong to?  shift 0: prl + e; shift 1: only e == 1 is valid (row r0l); else none
    const int rowv = pair_ok ? prl + e : prl;
    const bool valid = pair_ok ? (sh == 0 || (sh == 1 && e == 1)) : (e == 0 && r0l < f);
    if (valid && rowv >= rbeg && rowv < f) {
      const int64_t gcb = P.bigcol_base[s];
      double w[R];
      if (b == 0) fwd_gather<R>(P, gcb, col0, k, rowv, w);
      else {
#pragma unroll
        for (int r = 0; r < R; ++r) w[r] = ver ? ld_agent(fwd_slot<R>(P, s, col0, k, rowv, r)) : *fwd_slot<R>(P, s, col0, k, rowv, r);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double sum = 0.0;
#pragma unroll
        for (int h = 0; h < NG; ++h) sum += part[((size_t)h * R + r) * ROWS + tid];
        if (ver) st_agent(fwd_slot<R>(P, s, col0, k, rowv, r), w[r] - sum); else *fwd_slot<R>(P, s, col0, k, rowv, r) = w[r] - sum;
      }
    }
  }
  if (ver) flow_publish(ver, vt0, vnt, vbase + (unsigned long long)b + 1ull);
}

template <int R, int ROWS = kUpdRows>
__global__ __launch_bounds__(256) void k_fwd_upd(DevPlan P, const int* __restrict__ list, int b) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  fwd_upd_body<R, ROWS>(P, list[blockIdx.y], b, (int)blockIdx.x, sm, nullptr, 0ull);
}
// one launch per block column b of the wide fronts of a level: workgroups [0, ny * kCS) are the partial block products (row tile
// x % ny, column split x / ny), the others the panel rows below the block, waiting for their front's partials
template <int R>
__global__ __launch_bounds__(256) void k_fwd_wide_fused(DevPlan P, const int* __restrict__ list, int b, int ny, unsigned long long* __restrict__ counters, unsigned long long epoch) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int s = list[blockIdx.y];
  const int x = (int)blockIdx.x;
  if (x < ny * kCS) {
    if (fwd_y_body<R, true>(P, s, b, x % ny, x / ny)) front_arrive64(counters + s, epoch);
  } else {
    fwd_upd_body<R>(P, s, b, x - ny * kCS, sm, counters + s, epoch);
  }
}

// Flow launch, forward: EVERY block of the wide fronts of a level in one launch.  blockIdx.x = front, blockIdx.y = task; the tasks
// are laid out block by block -- the partial block products of block b (row tile x column split), then the panel rows below it:
// 32 workgroups of 32 rows (the rows the NEXT block product waits for: short tasks), then workgroups of 64 rows -- so that every
// task depends on tasks with SMALLER indices only (workgroups are dispatched in index order: whatever a waiting workgroup waits
// for is resident or done).  The block product of block b + 1 starts when the 32-row tiles of its rows carry b + 1 updates, while
// the far rows of block b (and b - 1, ...) are still streaming their panels: the chain of a front is block product -> near rows,
// the bulk of L streams beside it.
// MEASURED SLOWER than the two launches per block (k_fwd_y, k_fwd_upd) it was meant to replace -- 65 us per level instead of 53 at
// S-metric (650 / 534 us per solve) -- and therefore off (OKKT_SOLVE_FLOW=1 enables it; tests keep it correct): a dependent kernel
// boundary costs 1.5 - 2 us on this GPU, a hand-off through a flag with an agent-scope payload (drain, atomics, poll, payload
// loads that bypass the L1) 3 - 4 us, and the chain of a block -- product, then the 32-row panel tasks, each a 256 KB stream in four
// dependent rounds -- is the same length in both forms; what the flow launch adds is spinning workgroups beside the working ones.
constexpr int kFlowBlocks = 96;        // blocks of a wide front per flow launch (fronts with more pivot columns take the two-launch path)
constexpr int kFlowNear = 32;          // 32-row workgroups behind a block
struct SweepFlow {
  int nblk;
  int base[kFlowBlocks + 1];           // first task of block b
  short nprod[kFlowBlocks];            // block-product tasks of block b (tiles x kCS)
};
template <int R>
__global__ __launch_bounds__(256) void k_fwd_wide_flow(DevPlan P, const int* __restrict__ list, SweepFlow T, unsigned long long* __restrict__ counters, unsigned long long epoch) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int s = list[blockIdx.x];
  const int t = (int)blockIdx.y;
  int b = 0;
  while (b + 1 < T.nblk && t >= T.base[b + 1]) ++b;
  const int idx = t - T.base[b];
  const int k = P.sn_col0[s + 1] - P.sn_col0[s];
  int c0, kb, ld; int64_t off;
  xblock(k, b, c0, kb, ld, off);
  if (kb <= 0) return;
  unsigned long long* ver = P.sver + P.sver_pos[s];
  const unsigned long long vbase = epoch << 12;
  const int nprod = T.nprod[b];
  // the block's own arrival counter (behind the tile words): products of later blocks that are all zero arrive at once
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  unsigned long long* ycnt = ver + ((f + 31) >> 5) + 2 + b;
  if (idx < nprod) {
    const int ny = nprod / kCS;
    if (fwd_y_body<R, true>(P, s, b, idx % ny, idx / ny, ver, vbase + (unsigned long long)b)) front_arrive64(ycnt, epoch);
    return;
  }
  const int j = idx - nprod;
  const int t0 = (c0 + kb) >> 5;                                         // the 32-row tile that holds the first row below the block
  const long long yexp = (long long)((kb + 63) / 64) * kCS;
  if (j < kFlowNear) fwd_upd_body<R, 32>(P, s, b, j, sm, ycnt, epoch, ver, vbase, 32 * (t0 + j), j == 0, yexp);
  else fwd_upd_body<R, 64>(P, s, b, j, sm, ycnt, epoch, ver, vbase, 32 * (t0 + kFlowNear) + 64 * (j - kFlowNear), 0, yexp);
}

// ------------------------------------------------------------------------------------------------------------------
// big fronts, backward
// ------------------------------------------------------------------------------------------------------------------
// rows below the pivot block: z[c] -= sum_{r >= k} L[r, c] x[rows[r]]   (wave per column, eight loads in flight); z lives in
// zwork (forward result), the ancestors' solution in xwork
template <int R, bool SC1>
__device__ __forceinline__ void bwd_pre_body(const DevPlan& P, int s, int xblk) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int c = xblk * 4 + wv;
  if (c >= k || f == k) return;
  const int* rows = P.rows + P.row_ptr[s];
  const double* col = P.arena + P.front_pos[s] + (size_t)c * f;
  double acc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) acc[q] = 0.0;
  int r = k + lane;
  for (; r + 7 * 64 < f; r += 8 * 64) {
    int ri[8];
    double lv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { ri[u] = rows[r + 64 * u]; lv[u] = col[r + 64 * u]; }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      double xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = P.xwork[(size_t)q * P.xw_stride + ri[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[q] += lv[u] * xv[u];
    }
  }
  for (; r < f; r += 64) {
    const double l = col[r];
    const int g = rows[r];
#pragma unroll
    for (int q = 0; q < R; ++q) acc[q] += l * P.xwork[(size_t)q * P.xw_stride + g];
  }
#pragma unroll
  for (int q = 0; q < R; ++q) {
    double a = acc[q];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if (lane == 0) {
      double* zp = &P.zwork[(size_t)q * P.xw_stride + col0 + c];
      if (SC1) st_agent(zp, *zp - a); else *zp -= a;
    }
  }
}
template <int R>
__global__ __launch_bounds__(256) void k_bwd_pre(DevPlan P, const int* __restrict__ list) {
  bwd_pre_body<R, false>(P, list[blockIdx.y], (int)blockIdx.x);
}

// fronts with k <= 128: x_K = X' t (one workgroup per front; column c of X is contiguous).  counter != NULL (fused launch): the
// products of the rows below the pivot block arrive from `expected` workgroups of the same launch; X is loaded before the wait
template <int R>
__device__ __forceinline__ void bwd_thin_body(const DevPlan& P, int s, int NB, int* counter, double (*tk)[128]) {
  const int tid = threadIdx.x;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const double* X = P.invl + P.invl_pos[s];
  const int c = tid >> 1, h = tid & 1;
  const double* xcol = X + (size_t)min(c, NB - 1) * NB;
  double v[64];
#pragma unroll
  for (int q = 0; q < 64; ++q) v[q] = xcol[min(h * 64 + q, NB - 1)];
  if (counter) {
    const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
    const int expected = f > k ? (k + 3) / 4 : 0;
    front_wait(counter, expected, P.counters + 5);
    if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody has arrived: ready for the next solve
  }
  if (tid < 128)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double* zp = &P.zwork[(size_t)r * P.xw_stride + col0 + min(tid, max(k - 1, 0))];
      tk[r][tid] = tid < k ? (counter ? ld_agent(zp) : *zp) : 0.0;
    }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) a += (h * 64 + q < NB) ? v[q] * tk[r][h * 64 + q] : 0.0;
    a += __shfl_xor(a, 1, 64);
    if (h == 0 && c < k) P.xwork[(size_t)r * P.xw_stride + col0 + c] = a;
  }
}
template <int R>
__global__ __launch_bounds__(256) void k_bwd_thin(DevPlan P, const int* __restrict__ list, int NB) {
  __shared__ double tk[R][128];
  bwd_thin_body<R>(P, list[blockIdx.x], NB, nullptr, tk);
}
// one launch per level: workgroups 0 .. npre - 1 of a front (four columns each) fold the rows below the pivot block into z and
// arrive on the front's counter; the last workgroup waits for them and finishes the pivot block
template <int R>
__global__ __launch_bounds__(256) void k_bwd_thin_fused(DevPlan P, const int* __restrict__ list, int NB, int* __restrict__ counters) {
  __shared__ double tk[R][128];
  const int s = list[blockIdx.y];
  if (blockIdx.x + 1 < gridDim.x) {
    const int k = P.sn_col0[s + 1] - P.sn_col0[s];
    const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
    if ((int)blockIdx.x * 4 >= k || f == k) return;         // nothing to add for this front: not counted either
    bwd_pre_body<R, true>(P, s, (int)blockIdx.x);
    front_signal_add(counters + s);
  } else {
    bwd_thin_body<R>(P, s, NB, counters + s, tk);
  }
}

// wide fronts, block b: partial products of x_b = X_b' z_b.  Workgroup (j, front, rq): columns [64 j, +64), rows
// [rq * kSB / kCS, +kSB / kCS); column i of X_b is zero above row i.  A wave takes 16 columns, four at a time.
template <int R, bool AG>
__device__ __forceinline__ bool bwd_x_body(const DevPlan& P, int s, int b, int bx, int rq) {
  constexpr int RW = kSB / kCS;
  __shared__ double zj[R][RW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  int c0, kb, ld; int64_t off;
  xblock(k, b, c0, kb, ld, off);
  if (kb <= 0) return false;
  if (bx * 64 >= kb) return false;
  double* xp = P.ypart + P.ypart_pos[s] + (size_t)rq * kSB;
  const int i0 = bx * 64 + wv * 16;
  const int q0 = rq * RW;
  if (q0 >= kb || q0 + RW - 1 < bx * 64) {        // nothing of this row range reaches these columns
    if (lane < 16 && i0 + lane < kb)
#pragma unroll
      for (int r = 0; r < R; ++r) { if (AG) st_agent(&xp[(size_t)r * kCS * kSB + i0 + lane], 0.0); else xp[(size_t)r * kCS * kSB + i0 + lane] = 0.0; }
    return true;
  }
  const int kq = min(RW, kb - q0);
  const double* X = P.xinv + P.xinv_pos[s] + off + q0;
  for (int p = tid; p < RW; p += 256)
#pragma unroll
    for (int r = 0; r < R; ++r) zj[r][p] = p < kq ? P.zwork[(size_t)r * P.xw_stride + col0 + c0 + q0 + p] : 0.0;
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < 16; g += 4) {
    double v[4][RW / 64];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* xc = X + (size_t)min(i0 + g + q, kb - 1) * ld;
#pragma unroll
      for (int t = 0; t < RW / 64; ++t) v[q][t] = xc[min(lane + 64 * t, kq - 1)];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double a = 0.0;
#pragma unroll
        for (int t = 0; t < RW / 64; ++t) a += (lane + 64 * t < kq) ? v[q][t] * zj[r][lane + 64 * t] : 0.0;
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0 && i0 + g + q < kb) { if (AG) st_agent(&xp[(size_t)r * kCS * kSB + i0 + g + q], a); else xp[(size_t)r * kCS * kSB + i0 + g + q] = a; }
      }
    }
  }
  return true;
}
template <int R>
__global__ __launch_bounds__(256) void k_bwd_x(DevPlan P, const int* __restrict__ list, int b) {
  if (P.solve_mid) { const int s = list[blockIdx.y]; if (P.sn_col0[s + 1] - P.sn_col0[s] <= P.solve_mid) return; }      // k_bwd_mid's fronts
  (void)bwd_x_body<R, false>(P, list[blockIdx.y], b, (int)blockIdx.x, (int)blockIdx.z);
}

// wide fronts, block b: x_b = sum of the partials; columns c < c0: z[c] -= sum_p L[c0 + p][c] x_b[p] (wave: 16 columns, four
// at a time, lanes along the rows of the block); the workgroup behind the last column group stores x_b itself
// COLS = columns per workgroup: 64 (a wave takes 16) or 16 (a wave takes 4) for launches of few workgroups (the early blocks of a front)
template <int R, int COLS = 64>
__device__ __forceinline__ void bwd_upd_body(const DevPlan& P, int s, int b, int bx, double* sm, const unsigned long long* counter, unsigned long long epoch) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;       // sm: x[R][kSB]
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  int c0, kb, ld; int64_t off;
  xblock(k, b, c0, kb, ld, off);
  if (kb <= 0) return;
  const int nupd = (c0 + COLS - 1) / COLS;        // workgroups that update columns; the next one copies x_b
  if (bx > nupd) return;
  if (counter) front_wait64(counter, epoch, ((kb + 63) / 64) * kCS, P.counters + 5);
  const double* xp = P.ypart + P.ypart_pos[s];
  for (int p = tid; p < kSB; p += 256) {
    const int pc = min(p, kb - 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double x = counter ? sum_splits_agent(xp + (size_t)r * kCS * kSB + pc) : sum_splits(xp + (size_t)r * kCS * kSB + pc);
      sm[r * kSB + p] = p < kb ? x : 0.0;
    }
  }
  __syncthreads();
  if (bx == nupd) {
    for (int p = tid; p < kb; p += 256)
#pragma unroll
      for (int r = 0; r < R; ++r) P.xwork[(size_t)r * P.xw_stride + col0 + c0 + p] = sm[r * kSB + p];
    return;
  }
  const double* Lrow = P.arena + P.front_pos[s] + c0;
#pragma unroll 1
  for (int g = 0; g < COLS / 4; g += 2) {
    const int cb = bx * COLS + wv * (COLS / 4) + g;
    if (cb >= c0) break;
    double acc[2][R];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[q][r] = 0.0;
    for (int t0 = 0; t0 < kb; t0 += 64 * 16) {                 // 16 loads per lane and column in flight
      double v[2][16];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const double* col = Lrow + (size_t)min(cb + q, c0 - 1) * f;
#pragma unroll
        for (int t = 0; t < 16; ++t) v[q][t] = col[min(t0 + lane + 64 * t, kb - 1)];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int p = t0 + lane + 64 * t;
        if (p < kb) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const double x = sm[r * kSB + p];
            acc[0][r] += v[0][t] * x;
            acc[1][r] += v[1][t] * x;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double a = acc[q][r];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0 && cb + q < c0) P.zwork[(size_t)r * P.xw_stride + col0 + cb + q] -= a;
      }
  }
}
template <int R, int COLS = 64>
__global__ __launch_bounds__(256) void k_bwd_upd(DevPlan P, const int* __restrict__ list, int b) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  bwd_upd_body<R, COLS>(P, list[blockIdx.y], b, (int)blockIdx.x, sm, nullptr, 0ull);
}
// one launch per block column b, backward: workgroups [0, nx * kCS) are the partial products of x_b = X_b' z_b (column tile
// x % nx, row split x / nx), the others update the columns to the left / store x_b once their front's partials have arrived
template <int R>
__global__ __launch_bounds__(256) void k_bwd_wide_fused(DevPlan P, const int* __restrict__ list, int b, int nx, unsigned long long* __restrict__ counters, unsigned long long epoch) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int s = list[blockIdx.y];
  const int x = (int)blockIdx.x;
  if (x < nx * kCS) {
    if (bwd_x_body<R, true>(P, s, b, x % nx, x / nx)) front_arrive64(counters + s, epoch);
  } else {
    bwd_upd_body<R>(P, s, b, x - nx * kCS, sm, counters + s, epoch);
  }
}

// ---- fronts of NB + 1 .. kSB pivot columns: the pivot block by BLOCK SUBSTITUTION in 64-column steps --------------------------------
// An explicit inverse is as accurate as substitution for blocks of up to 64 columns and 5 - 40 x worse from 128 columns on (measured on
// BASELINE configs 3 and 5 with the HIP factor: scripts/solve_emulation.py).  These fronts have ONE block product
// per sweep (k <= kSB): one workgroup per front takes its place -- y_s = X64_s w_s with the 64 x 64 diagonal blocks of the 128-column
// inverses the factorisation leaves in `invl`, then the rest of the pivot block's rows minus L y_s, step by step -- and hands the result
// over where the block product's partial vectors go (split 0; the other splits zero), so that k_fwd_upd / k_bwd_upd stay as they are.
// The fronts of more than kSB pivot columns (launch chains of kSB-column inverses) are not touched.
constexpr int kMidThreads = 1024;
// the 64 x 64 diagonal block of a 128-column inverse: four values per thread, requested a step ahead (registers), stored with leading dimension 65
__device__ __forceinline__ void mid_load_x(const double* Xb, int o, int NB, double (&xr)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + kMidThreads * u, i = idx & 63, j = idx >> 6;
    xr[u] = Xb[(o + i) + (size_t)(o + j) * NB];
  }
}
__device__ __forceinline__ void mid_store_x(const double (&xr)[4], double* xb) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + kMidThreads * u, i = idx & 63, j = idx >> 6;
    xb[i + j * 65] = xr[u];
  }
}
// Three barriers per step: the 64 x 64 block of the NEXT step goes into the other of two LDS buffers while this step's rows are updated,
// and the four column quarters of a row pair sit in neighbouring lanes and meet by two shuffles.  (A form with one buffer, the quarters'
// partial sums in LDS and five barriers per step took the same 17 us per level and sweep; so did 256 and 512 threads per workgroup:
// what a step costs, about 2 us, is its chain of dependent LDS round trips, not the barriers.)
template <int R>
__global__ __launch_bounds__(kMidThreads) void k_fwd_mid(DevPlan P, const int* __restrict__ list, int NB) {
  extern __shared__ __attribute__((aligned(16))) double sm[];      // w[R][kSB], then the product's partial sums [16][R][64]
  __shared__ double xb[2][64 * 65], ys[R][64];
  double (*part)[R][64] = (double (*)[R][64])(sm + (size_t)R * kSB);
  const int s = list[blockIdx.x];
  const int tid = threadIdx.x;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  if (k > P.solve_mid) return;      // (a pivot block of up to NB columns: a thin front routed through the wide fronts' launches)
  const double* F = P.arena + P.front_pos[s];
  const double* X = P.invl + P.invl_pos[s];
  const int q = tid & 3, pr = tid >> 2;      // column quarter (neighbouring lanes) and row pair of the row update
  // the L entries of a step's row update: rows (p, p + 1) below the step, columns 16 q .. 16 q + 15 of it; requested a step ahead (they do
  // not depend on the step's solution), so that a step waits for LDS only
  auto load_panel = [&](int c0, int pbase, d2_t (&v)[16]) {
    const int nb = min(64, k - c0);
    const int p = pbase + 2 * pr;
    if (c0 < k && p < k) {
      const bool two = p + 1 < k;
      const double* Lr = F + (size_t)(c0 + 16 * q) * f + (two ? p : p - 1);      // a lone last row is the second value of the pair before it
#pragma unroll
      for (int u = 0; u < 16; ++u) __builtin_memcpy(&v[u], Lr + (size_t)min(u, max(nb - 1 - 16 * q, 0)) * f, 16);
    }
  };
  double xr[4];
  d2_t pv[16];
  mid_load_x(X, 0, NB, xr);
  load_panel(0, 64, pv);
  const int64_t gcb = P.bigcol_base[s];
  for (int p = tid; p < k; p += kMidThreads) {
    double t[R];
    fwd_gather<R>(P, gcb, col0, k, p, t);
#pragma unroll
    for (int r = 0; r < R; ++r) sm[r * kSB + p] = t[r];
  }
  mid_store_x(xr, xb[0]);
  if (64 < k) mid_load_x(X + (size_t)(64 >> 7) * NB * NB, 64 & 127, NB, xr);
  __syncthreads();
  for (int c0 = 0, st = 0; c0 < k; c0 += 64, ++st) {
    const int nb = min(64, k - c0);
    const double* xc = xb[st & 1];
    {
      // y_s = X64 w_s: row i = tid & 63, sixteen column groups of four
      const int i = tid & 63, g = tid >> 6;
      double a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = 0.0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * g + jj;
        const double x = (j <= i && i < nb) ? xc[i + j * 65] : 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] += x * sm[r * kSB + min(c0 + j, k - 1)];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) part[g][r][i] = a[r];
    }
    __syncthreads();
    if (tid < 64)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double v = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += part[g][r][tid];
        ys[r][tid] = tid < nb ? v : 0.0;
        if (tid < nb) sm[r * kSB + c0 + tid] = v;
      }
    __syncthreads();
    // the rows of the pivot block below this step
    for (int pbase = c0 + 64; pbase < k; pbase += 512) {
      if (pbase > c0 + 64) load_panel(c0, pbase, pv);      // (only the first 512 rows were requested ahead)
      const int p = pbase + 2 * pr;
      const bool have = p < k, two = p + 1 < k;
      double a0[R], a1[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { a0[r] = 0.0; a1[r] = 0.0; }
      if (have) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (16 * q + u < nb) {
#pragma unroll
            for (int r = 0; r < R; ++r) { const double y = ys[r][16 * q + u]; a0[r] += pv[u][0] * y; a1[r] += pv[u][1] * y; }
          }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        a0[r] += __shfl_xor(a0[r], 1, 64); a1[r] += __shfl_xor(a1[r], 1, 64);
        a0[r] += __shfl_xor(a0[r], 2, 64); a1[r] += __shfl_xor(a1[r], 2, 64);
        if (have && q == 0) {
          if (two) { sm[r * kSB + p] -= a0[r]; sm[r * kSB + p + 1] -= a1[r]; }
          else sm[r * kSB + p] -= a1[r];
        }
      }
    }
    if (c0 + 64 < k) {
      mid_store_x(xr, xb[(st + 1) & 1]);
      if (c0 + 128 < k) mid_load_x(X + (size_t)((c0 + 128) >> 7) * NB * NB, (c0 + 128) & 127, NB, xr);
      load_panel(c0 + 64, c0 + 128, pv);      // the next step's first 512 rows
    }
    __syncthreads();
  }
  double* yp = P.ypart + P.ypart_pos[s];
  for (int p = tid; p < k; p += kMidThreads)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      yp[(size_t)r * kCS * kSB + p] = sm[r * kSB + p];
#pragma unroll
      for (int q2 = 1; q2 < kCS; ++q2) yp[(size_t)r * kCS * kSB + (size_t)q2 * kSB + p] = 0.0;
    }
}
// backward: x_K = L_KK^-T z_K from the last 64 columns to the first (z_K already holds the products with the rows below: k_bwd_pre).
// The columns before a step are updated with the step's solution, t[c] -= sum_p L[c0 + p][c] x[p]: the 64 rows of a step are contiguous in a
// column, so HALF A WAVE takes a column (32 lanes x one row pair = one 512-byte run), 32 columns per pass of the workgroup; the row sums
// meet by five shuffles.  (A thread per column with sixteen scalar loads each -- 16 000 uncoalesced requests per step -- took 10 us per step.)
template <int R>
__global__ __launch_bounds__(kMidThreads) void k_bwd_mid(DevPlan P, const int* __restrict__ list, int NB) {
  extern __shared__ __attribute__((aligned(16))) double sm[];      // t[R][kSB], then the product's partial sums [16][R][64]
  __shared__ double xb[2][64 * 65], xs[R][64];
  double (*part)[R][64] = (double (*)[R][64])(sm + (size_t)R * kSB);
  const int s = list[blockIdx.x];
  const int tid = threadIdx.x;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  if (k > P.solve_mid) return;      // (a pivot block of up to NB columns: a thin front routed through the wide fronts' launches)
  const double* F = P.arena + P.front_pos[s];
  const double* X = P.invl + P.invl_pos[s];
  const int clast = ((k - 1) >> 6) << 6;
  const int hl = tid & 31, hw = tid >> 5;      // lane of the half-wave (row pair 2 hl), half-wave 0 .. 31
  constexpr int NPF = 8;                        // columns per thread requested a step ahead: 32 half-waves x 8 = the first 256 columns
  auto load_cols = [&](int c0, int cfirst, d2_t (&v)[NPF]) {
    if (c0 < 0) return;
    const int nb = min(64, k - c0);
    // a short last step: clamped pair (masked when used).  A one-row step (nb = 1) takes the pair that ENDS in its row, (c0 - 1, c0):
    // the pair starting there would reach the row below the pivot block -- or, in a front without such rows, the never-written entry
    // above the next column's diagonal, whose NaN survives the multiplication by the zero that masks it (fuzz, round 5).  Row c0 - 1 is
    // on or below the diagonal of every column c < c0: written, finite.
    const int pp = min(2 * hl, nb - 2);
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
      const int c = cfirst + hw + 32 * u;
      if (c < c0) __builtin_memcpy(&v[u], F + (size_t)c * f + c0 + pp, 16);
    }
  };
  double xr[4];
  d2_t cv[NPF];
  mid_load_x(X + (size_t)(clast >> 7) * NB * NB, clast & 127, NB, xr);
  load_cols(clast, 0, cv);
  for (int p = tid; p < k; p += kMidThreads)
#pragma unroll
    for (int r = 0; r < R; ++r) sm[r * kSB + p] = P.zwork[(size_t)r * P.xw_stride + col0 + p];
  mid_store_x(xr, xb[0]);
  if (clast >= 64) mid_load_x(X + (size_t)((clast - 64) >> 7) * NB * NB, (clast - 64) & 127, NB, xr);
  __syncthreads();
  for (int c0 = clast, st = 0; c0 >= 0; c0 -= 64, ++st) {
    const int nb = min(64, k - c0);
    const double* xc = xb[st & 1];
    {
      // x_s = X64' t_s: column j = tid & 63, sixteen row groups of four
      const int j = tid & 63, g = tid >> 6;
      double a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = 0.0;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * g + ii;
        const double x = (i >= j && i < nb) ? xc[i + j * 65] : 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] += x * sm[r * kSB + min(c0 + i, k - 1)];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) part[g][r][j] = a[r];
    }
    __syncthreads();
    if (tid < 64)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double v = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += part[g][r][tid];
        xs[r][tid] = tid < nb ? v : 0.0;
        if (tid < nb) sm[r * kSB + c0 + tid] = v;
      }
    __syncthreads();
    {
      // the step's solution at this lane's row pair (zero beyond a short last step; an odd nb: its last row is the SECOND value of the clamped pair)
      const int pp = min(2 * hl, nb - 2);      // as in load_cols (-1 for a one-row step: its row is the second value, the case below)
      double x0[R], x1[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        x0[r] = (pp == 2 * hl && pp < nb) ? xs[r][pp] : 0.0;
        x1[r] = (pp == 2 * hl && pp + 1 < nb) ? xs[r][pp + 1] : 0.0;
      }
      if (pp != 2 * hl && 2 * hl == nb - 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) { x0[r] = 0.0; x1[r] = xs[r][nb - 1]; }
      }
      for (int cfirst = 0; cfirst < c0; cfirst += 32 * NPF) {
        if (cfirst > 0) load_cols(c0, cfirst, cv);
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
          const int c = cfirst + hw + 32 * u;
          if (c < c0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
              double a = cv[u][0] * x0[r] + cv[u][1] * x1[r];
              for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
              if (hl == 0) sm[r * kSB + c] -= a;
            }
          }
        }
      }
    }
    if (c0 >= 64) {
      mid_store_x(xr, xb[(st + 1) & 1]);
      if (c0 >= 128) mid_load_x(X + (size_t)((c0 - 128) >> 7) * NB * NB, (c0 - 128) & 127, NB, xr);
      load_cols(c0 - 64, 0, cv);      // the next step's first 256 columns
    }
    __syncthreads();
  }
  // x_K itself (what k_bwd_upd would copy out of the partial vectors: a level whose wide fronts all come through here skips that launch),
  // and the partial vectors for the levels that still run it
  double* xp = P.ypart + P.ypart_pos[s];
  for (int p = tid; p < k; p += kMidThreads)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      P.xwork[(size_t)r * P.xw_stride + col0 + p] = sm[r * kSB + p];
      xp[(size_t)r * kCS * kSB + p] = sm[r * kSB + p];
#pragma unroll
      for (int q2 = 1; q2 < kCS; ++q2) xp[(size_t)r * kCS * kSB + (size_t)q2 * kSB + p] = 0.0;
    }
}

template <int R>
__global__ void k_permute_in_r(int n, int nr, int64_t stride_in, const int* __restrict__ perm, const double* __restrict__ rhs, double* __restrict__ x, int64_t xs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = perm[i];
#pragma unroll
  for (int q = 0; q < R; ++q) x[(size_t)q * xs + i] = q < nr ? rhs[(size_t)q * stride_in + g] : 0.0;
}
template <int R>
__global__ void k_permute_out_r(int n, int nr, int64_t stride_out, const int* __restrict__ perm, const double* __restrict__ x, int64_t xs, double* __restrict__ sol, int accumulate,
                                const unsigned long long* __restrict__ tmo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = perm[i];
  // an in-launch hand-off of this solve (or of the factorisation behind it) ran into its bound: no solution rather than a wrong one
  const bool lost = __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
#pragma unroll
  for (int q = 0; q < R; ++q)
    if (q < nr) {
      double* d = sol + (size_t)q * stride_out + g;
      const double v = lost ? __builtin_nan("") : x[(size_t)q * xs + i];
      *d = accumulate ? *d + v : v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
namespace {
template <typename T>
std::string up(Numeric& N, const std::vector<T>& v, T** out) {
  *out = nullptr;
  void* p = nullptr;
  OKKT_HIP_TRY(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  N.allocations.push_back(p);
  if (!v.empty()) OKKT_HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return "";
}
std::string dz(Numeric& N, size_t count, double** out) {
  *out = nullptr;
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(double);
  OKKT_HIP_TRY(hipMalloc(&p, bytes));
  N.allocations.push_back(p);
  OKKT_HIP_TRY(hipMemset(p, 0, bytes));
  OKKT_HIP_TRY(hipStreamSynchronize(nullptr));
  *out = (double*)p;
  return "";
}
}  // namespace

// per level: the big fronts split by the width of the pivot block (thin: k <= NB, wide: k > NB), device lists + host segments
std::string solve_setup(const Symbolic& S, Numeric& N) {
  DevPlan& d = N.d;
  // the widest pivot block that goes through block substitution (0: none).  Measured (S-C5 / S-metric, forward error of the worse of two
  // vectors and solve time; the CPU restatement: 6.9e-10 / 9.7e-9): none 1.2e-8, 1.04 ms / 2.0e-8, 1.54 ms; 256: 1.2e-9, 1.26 / 9.2e-9, 1.75;
  // 384: 9.3e-10, 1.26 / 8.4e-9, 1.76; 512: 9.3e-10, 1.26 / 1.3e-8, 1.89; 1024: 1.6e-9, 1.78 / 9.6e-9, 2.05 -- the many fronts of
  // 129 .. 256 columns carry the error, the few wide ones the time
  d.solve_mid = std::min(getenv("OKKT_SOLVE_MID") ? atoi(getenv("OKKT_SOLVE_MID")) : 384, kSB);
  if (kSolveBlock != 1024) d.solve_mid = 0;      // the mid kernels' LDS (static 66.5 KB + R * (kSB + 1024) doubles) is sized for 1024-column blocks: a 2048-column build keeps the explicit inverses (advisor, round 5)
  if (N.solve_flow || N.solve_fuse_wide_max > 0 || N.nb != 128) d.solve_mid = 0;      // (the flow and fused-wide experiments have their own block products with the explicit inverses; the 64-column inverses are read out of 128-column blocks)
  const int ns = S.nsuper;
  std::vector<int> ssched;
  std::vector<int64_t> xinv_pos(ns, -1), ypart_pos(ns, -1), ythin_pos(ns, -1), sver_pos(ns, -1);
  int64_t xtot = 0, ytot = 0, ttot = 0, vtot = 0;
  static const int route_thin = getenv("OKKT_SOLVE_ROUTE_THIN") ? atoi(getenv("OKKT_SOLVE_ROUTE_THIN")) : 1;
  auto build = [&](const std::vector<LevelSchedule>& levels, std::vector<SolveLevel>& out) {
    out.assign(levels.size(), SolveLevel());
    for (size_t l = 0; l < levels.size(); ++l) {
      const Segment& g = levels[l].seg[3];
      SolveLevel& L = out[l];
      std::vector<int> thin, wide;
      for (int q = 0; q < g.cnt; ++q) {
        const int s = N.sched_host[g.off + q];
        (N.sn_k[s] <= N.nb ? thin : wide).push_back(s);
      }
      // A level that has wide fronts anyway runs the block-substitution launch for them (k_fwd_mid / k_bwd_mid, one workgroup per front) and
      // the panel launches behind it; its thin fronts (k <= NB) then cost a launch of their own per sweep on top -- 20 us of critical path
      // per level and sweep for a few hundred KB.  They go through the wide fronts' launches instead (round 6): a pivot block of up to NB
      // columns is one or two 64-column steps of the same substitution -- and more accurate than the product with its 128 x 128 inverse
      // (DESIGN.md section 5).  Levels without a wide front keep the fused thin launch (one launch instead of two).
      if (route_thin && d.solve_mid >= N.nb && !thin.empty() && !wide.empty()) { wide.insert(wide.end(), thin.begin(), thin.end()); thin.clear(); }
      L.thin_off = (int)ssched.size(); L.thin_cnt = (int)thin.size();
      for (int s : thin) { ssched.push_back(s); if (ythin_pos[s] < 0) { ythin_pos[s] = ttot; ttot += (int64_t)kMaxRhs * 128; } L.thin_maxf = std::max(L.thin_maxf, N.sn_f[s]); L.thin_maxk = std::max(L.thin_maxk, N.sn_k[s]); L.thin_maxr = std::max(L.thin_maxr, N.sn_f[s] - N.sn_k[s]); }
      L.wide_off = (int)ssched.size(); L.wide_cnt = (int)wide.size();
      for (int s : wide) {
        ssched.push_back(s);
        L.wide_maxf = std::max(L.wide_maxf, N.sn_f[s]); L.wide_maxk = std::max(L.wide_maxk, N.sn_k[s]); L.wide_mink = std::min(L.wide_mink, N.sn_k[s]);
        if (xinv_pos[s] < 0) {
          const int64_t k = N.sn_k[s];
          const int64_t nfull = k / kSB, last = k - nfull * kSB, lastp = (last + 127) / 128 * 128;
          xinv_pos[s] = xtot;
          xtot += nfull * (int64_t)kSB * kSB + lastp * lastp;
          ypart_pos[s] = ytot;
          ytot += (int64_t)kMaxRhs * kCS * kSB * (N.solve_flow ? (k + kSB - 1) / kSB : 1);      // one buffer of partial products (the flow experiment: one per block)
          sver_pos[s] = vtot;
          vtot += (N.sn_f[s] + 31) / 32 + 2 + (k + kSB - 1) / kSB;      // tile words, then one arrival counter per block
        }
      }
    }
  };
  build(N.levels, N.slevels);
  build(N.levels_top, N.slevels_top);
  std::string e;
  if (!(e = up(N, ssched, &d.ssched)).empty()) return e;
  if (!(e = up(N, xinv_pos, &d.xinv_pos)).empty()) return e;
  if (!(e = up(N, ypart_pos, &d.ypart_pos)).empty()) return e;
  if (!(e = up(N, ythin_pos, &d.ythin_pos)).empty()) return e;
  if (!(e = dz(N, (size_t)ttot, &d.ythin)).empty()) return e;
  if (!(e = dz(N, (size_t)xtot, &d.xinv)).empty()) return e;
  if (!(e = dz(N, (size_t)xtot, &d.xtmp)).empty()) return e;
  if (!(e = dz(N, (size_t)ytot, &d.ypart)).empty()) return e;
  if (!(e = up(N, sver_pos, &d.sver_pos)).empty()) return e;
  {
    double* rawv = nullptr;
    if (!(e = dz(N, (size_t)vtot + 1, &rawv)).empty()) return e;      // zero-filled once: the words carry the epoch of the launch that wrote them
    d.sver = (unsigned long long*)rawv;
  }
  {
    // hand-off words of the fused sweeps: a monotonic y flag and an arrival counter per supernode (zero-filled once)
    double* raw = nullptr;
    if (!(e = dz(N, (size_t)ns + 64, &raw)).empty()) return e;      // ns + 64 doubles = room for 2 x ns ints
    N.solve_flags = (int*)raw;
    N.solve_counters = (int*)raw + ns + 16;
    N.solve_epoch = 0;
    double* raw64 = nullptr;
    if (!(e = dz(N, (size_t)ns + 8, &raw64)).empty()) return e;
    N.solve_counters64 = (unsigned long long*)raw64;
    N.solve_epoch64 = 0;
  }
  for (const void* fn : {(const void*)k_fwd_wide_fused<1>, (const void*)k_fwd_wide_fused<2>, (const void*)k_fwd_wide_fused<4>, (const void*)k_bwd_wide_fused<1>,
                         (const void*)k_bwd_wide_fused<2>, (const void*)k_bwd_wide_fused<4>})
    OKKT_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096 - 16 * 1024));
  for (const void* fn : {(const void*)k_fwd_mid<1>, (const void*)k_fwd_mid<2>, (const void*)k_fwd_mid<4>, (const void*)k_bwd_mid<1>, (const void*)k_bwd_mid<2>, (const void*)k_bwd_mid<4>})
    OKKT_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  for (const void* fn : {(const void*)k_fwd_upd<1>, (const void*)k_fwd_upd<2>, (const void*)k_fwd_upd<4>, (const void*)k_bwd_upd<1>,
                         (const void*)k_bwd_upd<2>, (const void*)k_bwd_upd<4>, (const void*)k_fwd_upd<1, 32>, (const void*)k_fwd_upd<2, 32>, (const void*)k_fwd_upd<4, 32>, (const void*)k_bwd_upd<1, 16>, (const void*)k_bwd_upd<2, 16>, (const void*)k_bwd_upd<4, 16>, (const void*)k_fs_small<256, 1>, (const void*)k_fs_small<256, 2>,
                         (const void*)k_fs_small<256, 4>, (const void*)k_bs_small<256, 1>, (const void*)k_bs_small<256, 2>, (const void*)k_bs_small<256, 4>})
    OKKT_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
  return "";
}

// inverses of the diagonal blocks [b_lo, b_hi) of the wide fronts of one level (b_hi is clipped per front), on stream st:
// needs the NB x NB inverses of those block columns (k_big_invert) and the final L of the columns below them
std::string solve_invert_enqueue(Numeric& N, hipStream_t st, const SolveLevel& L, int b_lo, int b_hi) {
  if (!L.wide_cnt) return "";
  DevPlan P = N.d;
  const int* list = P.ssched + L.wide_off;
  const int nblk = (L.wide_maxk + kSB - 1) / kSB;
  b_hi = std::min(b_hi, nblk);
  if (b_lo >= b_hi) return "";
  const int nbk = b_hi - b_lo;
  const int maxkb = std::min(kSB, L.wide_maxk - b_lo * kSB);
  const int ldmax = std::min(kSB, (maxkb + 127) / 128 * 128);
  hipLaunchKernelGGL(k_xinv_init, dim3(ldmax / N.nb, 1, (unsigned)(L.wide_cnt * nbk)), dim3(256), 0, st, P, list, b_lo, nbk, N.nb);
  for (int sz = N.nb; sz < ldmax; sz *= 2) {
    const int npair = (ldmax + 2 * sz - 1) / (2 * sz);
    const int tiles = ((sz + 63) / 64) * ((sz + 63) / 64);
    hipLaunchKernelGGL(k_xinv_gemm<0>, dim3(tiles, npair, (unsigned)(L.wide_cnt * nbk)), dim3(256), 0, st, P, list, b_lo, nbk, sz);
    hipLaunchKernelGGL(k_xinv_gemm<1>, dim3(tiles, npair, (unsigned)(L.wide_cnt * nbk)), dim3(256), 0, st, P, list, b_lo, nbk, sz);
  }
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}

template <int R>
static std::string fwd_enqueue_r(Numeric& N, const std::vector<LevelSchedule>& levels, const std::vector<SolveLevel>& sl, hipStream_t st, int l_lo, int l_hi) {
  DevPlan P = N.d;
  const bool flow = &levels == &N.levels && N.flow_levels >= 2 && N.flow_flags && l_lo == 0 && l_hi >= N.flow_levels;
  for (size_t l = (size_t)l_lo; l < std::min((size_t)l_hi, levels.size()); ++l) {
    const LevelSchedule& L = levels[l];
    if (flow && l == 0) {     // the small-front tasks of the levels [0, flow_levels) in one launch
      const size_t lds = (size_t)R * N.flow_maxf * sizeof(double);
      int* flags = N.flow_flags + (size_t)N.d.nsuper;
      const int ep = ++N.flow_epoch;
      static const int drop = getenv("OKKT_DEBUG_DROP_HANDOFF") ? atoi(getenv("OKKT_DEBUG_DROP_HANDOFF")) : 0;   // tests: bit 1 = the forward sweep's waits never end
      const int wep = ep + ((drop & 2) ? (1 << 20) : 0);
      if (N.flow_maxf <= 32) hipLaunchKernelGGL((k_fs_small<64, R, true>), dim3(N.flow_cnt), dim3(64), lds, st, P, P.sched + N.flow_off, N.flow_maxf, flags, ep, wep);
      else hipLaunchKernelGGL((k_fs_small<256, R, true>), dim3(N.flow_cnt), dim3(256), lds, st, P, P.sched + N.flow_off, N.flow_maxf, flags, ep, wep);
      l = (size_t)N.flow_levels - 1;
      continue;
    }
    for (int c = 0; c < 3; ++c) {
      const Segment& g = L.seg[c];
      if (!g.cnt) continue;
      const size_t lds = (size_t)R * g.maxf * sizeof(double);
      if (c == 0) hipLaunchKernelGGL((k_fs_small<64, R>), dim3(g.cnt), dim3(64), lds, st, P, P.sched + g.off, g.maxf, (int*)nullptr, 0, 0);
      else hipLaunchKernelGGL((k_fs_small<256, R>), dim3(g.cnt), dim3(256), lds, st, P, P.sched + g.off, g.maxf, (int*)nullptr, 0, 0);
    }
    if (&levels == &N.levels && (size_t)l < N.inv_level_pending.size() && N.inv_level_pending[l]) {      // this level's block inverses (enqueued by the factorisation on the auxiliary stream)
      OKKT_HIP_TRY(hipStreamWaitEvent(st, N.inv_level_events[l], 0));
      N.inv_level_pending[l] = 0;
    }
    const SolveLevel& S = sl[l];
    if (S.thin_cnt && N.solve_fuse && N.solve_flags) {
      // one launch: workgroup 0 of a front computes y and raises its flag, the others apply the panel rows behind it
      hipLaunchKernelGGL(k_fwd_thin_fused<R>, dim3((S.thin_maxr + 127) / 128 + 1, S.thin_cnt), dim3(256), 0, st, P, P.ssched + S.thin_off, N.nb, N.solve_flags, N.solve_epoch);
    } else if (S.thin_cnt) {
      hipLaunchKernelGGL(k_fwd_thin_y<R>, dim3(S.thin_cnt), dim3(256), 0, st, P, P.ssched + S.thin_off, N.nb);
      if (S.thin_maxr > 0) hipLaunchKernelGGL(k_fwd_thin_upd<R>, dim3((S.thin_maxr + 127) / 128, S.thin_cnt), dim3(256), 0, st, P, P.ssched + S.thin_off);
    }
    if (S.wide_cnt) {
      const int* list = P.ssched + S.wide_off;
      const int nblk = (S.wide_maxk + kSB - 1) / kSB;
      const size_t lds = ((size_t)R * kSB + (size_t)8 * R * kUpdRows) * sizeof(double);
      bool flow_done = false;
      if (N.solve_flow && N.solve_counters64 && P.sver && nblk <= kFlowBlocks) {
        SweepFlow T;
        T.nblk = nblk;
        int64_t tot = 0;
        for (int b = 0; b < nblk; ++b) {
          const int kbmax = std::min(kSB, S.wide_maxk - b * kSB);
          const int rem = std::max(S.wide_maxf - b * kSB, 0);      // upper bound on the rows below the START of block b
          T.base[b] = (int)tot;
          T.nprod[b] = (short)(((kbmax + 63) / 64) * kCS);
          tot += T.nprod[b] + kFlowNear + std::max(0, (rem + 32 - 32 * kFlowNear + 63) / 64);
        }
        T.base[nblk] = (int)tot;
        if (tot <= 65535) {
          hipLaunchKernelGGL(k_fwd_wide_flow<R>, dim3(S.wide_cnt, (unsigned)tot), dim3(256), lds, st, P, list, T, N.solve_counters64, ++N.solve_epoch64);
          flow_done = true;
        }
      }
      for (int b = 0; b < nblk && !flow_done; ++b) {
        const int kbmax = std::min(kSB, S.wide_maxk - b * kSB);
        const int rem = std::max(S.wide_maxf - b * kSB, 0);        // upper bound on the rows below the start of block b (a narrower last block leaves more rows than maxf - (b + 1) kSB)
        const int ny = (kbmax + 63) / 64, nupd = std::max(1, (rem + kUpdRows - 1) / kUpdRows);
        if (N.solve_fuse && N.solve_counters64 && ny * kCS * S.wide_cnt <= N.solve_fuse_wide_max) {
          hipLaunchKernelGGL(k_fwd_wide_fused<R>, dim3(ny * kCS + nupd, S.wide_cnt), dim3(256), lds, st, P, list, b, ny, N.solve_counters64, ++N.solve_epoch64);
        } else {
          if (P.solve_mid && b == 0 && S.wide_mink <= P.solve_mid) hipLaunchKernelGGL(k_fwd_mid<R>, dim3(S.wide_cnt), dim3(kMidThreads), (size_t)R * (kSB + 16 * 64) * sizeof(double), st, P, list, N.nb);
          if (!P.solve_mid || S.wide_maxk > P.solve_mid) hipLaunchKernelGGL(k_fwd_y<R>, dim3(ny, S.wide_cnt, kCS), dim3(256), 0, st, P, list, b);
          // few workgroups of 64 rows (the later blocks of a front): 32 rows each, so that the launch reaches more CUs
          if (N.solve_split_small && nupd * S.wide_cnt < N.solve_split_small) hipLaunchKernelGGL((k_fwd_upd<R, 32>), dim3(std::max(1, (rem + 31) / 32), S.wide_cnt), dim3(256), lds, st, P, list, b);
          else hipLaunchKernelGGL(k_fwd_upd<R>, dim3(nupd, S.wide_cnt), dim3(256), lds, st, P, list, b);
        }
      }
    }
  }
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}

template <int R>
static std::string bwd_enqueue_r(Numeric& N, const std::vector<LevelSchedule>& levels, const std::vector<SolveLevel>& sl, hipStream_t st, int l_lo, int l_hi) {
  DevPlan P = N.d;
  const bool flow = &levels == &N.levels && N.flow_levels >= 2 && N.flow_flags && l_lo == 0 && l_hi >= N.flow_levels;
  for (int l = std::min(l_hi, (int)levels.size()) - 1; l >= l_lo; --l) {
    const LevelSchedule& L = levels[l];
    if (flow && l == N.flow_levels - 1) {     // the levels [0, flow_levels) in one launch, parents at the lower block indices
      const size_t lds = (size_t)R * N.flow_maxf * sizeof(double);
      int* flags = N.flow_flags + (size_t)2 * N.d.nsuper;
      const int ep = ++N.flow_epoch;
      if (N.flow_maxf <= 32) hipLaunchKernelGGL((k_bs_small<64, R, true>), dim3(N.flow_cnt), dim3(64), lds, st, P, P.sched + N.flow_off, N.flow_maxf, flags, ep);
      else hipLaunchKernelGGL((k_bs_small<256, R, true>), dim3(N.flow_cnt), dim3(256), lds, st, P, P.sched + N.flow_off, N.flow_maxf, flags, ep);
      break;
    }
    const SolveLevel& S = sl[l];
    if (S.wide_cnt) {
      const int* list = P.ssched + S.wide_off;
      if (S.wide_maxf > 0) hipLaunchKernelGGL(k_bwd_pre<R>, dim3((S.wide_maxk + 3) / 4, S.wide_cnt), dim3(256), 0, st, P, list);
      const int nblk = (S.wide_maxk + kSB - 1) / kSB;
      const size_t lds = (size_t)R * kSB * sizeof(double);
      for (int b = nblk - 1; b >= 0; --b) {
        const int kbmax = std::min(kSB, S.wide_maxk - b * kSB);
        const int nx = (kbmax + 63) / 64, nupd = (b * kSB + 63) / 64 + 1;
        if (N.solve_fuse && N.solve_counters64 && nx * kCS * S.wide_cnt <= N.solve_fuse_wide_max) {
          hipLaunchKernelGGL(k_bwd_wide_fused<R>, dim3(nx * kCS + nupd, S.wide_cnt), dim3(256), lds, st, P, list, b, nx, N.solve_counters64, ++N.solve_epoch64);
        } else {
          if (P.solve_mid && b == 0 && S.wide_mink <= P.solve_mid) hipLaunchKernelGGL(k_bwd_mid<R>, dim3(S.wide_cnt), dim3(kMidThreads), (size_t)R * (kSB + 16 * 64) * sizeof(double), st, P, list, N.nb);
          if (!P.solve_mid || S.wide_maxk > P.solve_mid) hipLaunchKernelGGL(k_bwd_x<R>, dim3(nx, S.wide_cnt, kCS), dim3(256), 0, st, P, list, b);
          if (P.solve_mid && S.wide_maxk <= P.solve_mid) continue;      // k_bwd_mid has stored x_K of every front of this level
          if (N.solve_split_small && nupd * S.wide_cnt < N.solve_split_small) hipLaunchKernelGGL((k_bwd_upd<R, 16>), dim3((b * kSB + 15) / 16 + 1, S.wide_cnt), dim3(256), lds, st, P, list, b);
          else hipLaunchKernelGGL(k_bwd_upd<R>, dim3(nupd, S.wide_cnt), dim3(256), lds, st, P, list, b);
        }
      }
    }
    if (S.thin_cnt && N.solve_fuse && N.solve_counters) {
      hipLaunchKernelGGL(k_bwd_thin_fused<R>, dim3((S.thin_maxk + 3) / 4 + 1, S.thin_cnt), dim3(256), 0, st, P, P.ssched + S.thin_off, N.nb, N.solve_counters);
    } else if (S.thin_cnt) {
      const int* list = P.ssched + S.thin_off;
      if (S.thin_maxr > 0) hipLaunchKernelGGL(k_bwd_pre<R>, dim3((S.thin_maxk + 3) / 4, S.thin_cnt), dim3(256), 0, st, P, list);
      hipLaunchKernelGGL(k_bwd_thin<R>, dim3(S.thin_cnt), dim3(256), 0, st, P, list, N.nb);
    }
    for (int c = 0; c < 3; ++c) {
      const Segment& g = L.seg[c];
      if (!g.cnt) continue;
      const size_t lds = (size_t)R * g.maxf * sizeof(double);
      if (c == 0) hipLaunchKernelGGL((k_bs_small<64, R>), dim3(g.cnt), dim3(64), lds, st, P, P.sched + g.off, g.maxf, (int*)nullptr, 0);
      else hipLaunchKernelGGL((k_bs_small<256, R>), dim3(g.cnt), dim3(256), lds, st, P, P.sched + g.off, g.maxf, (int*)nullptr, 0);
    }
  }
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}


static std::string sweep(Numeric& N, bool fwd, const std::vector<LevelSchedule>& levels, const std::vector<SolveLevel>& sl, hipStream_t st, int R,
                         int l_lo = 0, int l_hi = 1 << 30) {
  if (fwd) return R == 1 ? fwd_enqueue_r<1>(N, levels, sl, st, l_lo, l_hi) : (R == 2 ? fwd_enqueue_r<2>(N, levels, sl, st, l_lo, l_hi) : fwd_enqueue_r<4>(N, levels, sl, st, l_lo, l_hi));
  return R == 1 ? bwd_enqueue_r<1>(N, levels, sl, st, l_lo, l_hi) : (R == 2 ? bwd_enqueue_r<2>(N, levels, sl, st, l_lo, l_hi) : bwd_enqueue_r<4>(N, levels, sl, st, l_lo, l_hi));
}

// which = 0: the local subtrees (everything when the plan is not partitioned); which = 1: the top schedule of a partitioned plan
static std::string sweep_which(Numeric& N, bool fwd, int which, int R) {
  if (which != 0) return sweep(N, fwd, N.levels_top, N.slevels_top, N.stream, R);
  return sweep(N, fwd, N.levels, N.slevels, N.stream, R);
}

std::string solve_fwd_enqueue(Numeric& N, int which, int R) {
  if (which == 0 && N.inv_wait) { OKKT_HIP_TRY(hipStreamWaitEvent(N.stream, N.inv_event, 0)); N.inv_wait = false; }   // inversions started by the factorisation
  ++N.solve_epoch;          // the y flags of the fused forward launches are monotonic: one value per forward sweep
  return sweep_which(N, true, which, R);
}
std::string solve_bwd_enqueue(Numeric& N, int which, int R) {
  return sweep_which(N, false, which, R);
}

void solve_permute_in(const Numeric& N, const double* d_rhs, int64_t stride, int nr, int R) {
  const int n = N.d.n;
  if (!n) return;
  const dim3 g((n + 255) / 256), b(256);
  if (R == 1) hipLaunchKernelGGL(k_permute_in_r<1>, g, b, 0, N.stream, n, nr, stride, N.d.perm, d_rhs, N.d.xwork, N.d.xw_stride);
  else if (R == 2) hipLaunchKernelGGL(k_permute_in_r<2>, g, b, 0, N.stream, n, nr, stride, N.d.perm, d_rhs, N.d.xwork, N.d.xw_stride);
  else hipLaunchKernelGGL(k_permute_in_r<4>, g, b, 0, N.stream, n, nr, stride, N.d.perm, d_rhs, N.d.xwork, N.d.xw_stride);
}
void solve_permute_out(const Numeric& N, double* d_sol, int64_t stride, int nr, int R, bool accumulate) {
  const int n = N.d.n;
  if (!n) return;
  const dim3 g((n + 255) / 256), b(256);
  const int acc = accumulate ? 1 : 0;
  if (R == 1) hipLaunchKernelGGL(k_permute_out_r<1>, g, b, 0, N.stream, n, nr, stride, N.d.perm, N.d.xwork, N.d.xw_stride, d_sol, acc, N.d.counters + 5);
  else if (R == 2) hipLaunchKernelGGL(k_permute_out_r<2>, g, b, 0, N.stream, n, nr, stride, N.d.perm, N.d.xwork, N.d.xw_stride, d_sol, acc, N.d.counters + 5);
  else hipLaunchKernelGGL(k_permute_out_r<4>, g, b, 0, N.stream, n, nr, stride, N.d.perm, N.d.xwork, N.d.xw_stride, d_sol, acc, N.d.counters + 5);
}

}  // namespace okkt
