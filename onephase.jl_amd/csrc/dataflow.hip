// Dataflow factorisation of the big fronts of one level: ONE persistent launch instead of one launch per block column and kernel.
//
// Replaces, for the fronts above small_max rows, the per-step sequence k_big_diag2 -> k_big_trsm -> k_big_syrk of numeric.hip (the
// reference reaches the same arithmetic through CHOLMOD's numeric factorisation, /root/reference/src/linear_system_solvers/julia.jl:21-97).
// The level's fronts are cut into 128 x 128 tiles; dataflow_sched.cpp turns the blocked right-looking LDL^T into tasks D (diagonal
// tile), T (tile of the panel below it) and U (update of a trailing tile by one or two panels) and orders them in ONE queue.
// Every workgroup of this launch is a worker: pop the next task, wait until the tile states it depends on have been published
// (bounded spin on agent-scope loads), run the task's body, publish the new state of its tile.  No kernel boundaries and no
// stream events between the steps: the diagonal block of block column q + 1 starts as soon as ITS tile has received panel q,
// while the rest of panel q's update is still running on the other CUs.
//
// Hand-off protocol (MI355X guide, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): a producer writes every
// byte that another task reads with sc1 (write-through) stores, every storing wave drains its stores (s_waitcnt vmcnt(0)), the
// workgroup meets at a barrier, one lane publishes the tile state with an agent-scope store.  A consumer polls the state with
// agent-scope loads from one lane, that lane runs an agent-scope acquire (invalidates the CU's L1) and drains it, the workgroup
// meets at a barrier, and only then loads -- plain loads and LDS-DMA.
//
// The arithmetic per entry is the sequence of the per-step kernels (same bodies, same order of the panels per tile): the factor
// is bitwise the same, which is how the GPU tests pin this path (tests/test_gpu_dataflow.py).
#include "front_device.h"

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace okkt {

#define OKKT_HIP_TRY(expr)                                                         \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess)                                                         \
      return std::string(#expr) + ": " + hipGetErrorString(e__);                   \
  } while (0)

typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr int kDfThreadsC = 512;   // threads of a worker
#ifndef OKKT_DF_STAGES
#define OKKT_DF_STAGES 2
#endif
#ifndef OKKT_DF_KC
#define OKKT_DF_KC 32
#endif
#ifndef OKKT_DF_STAGGER
#define OKKT_DF_STAGGER 1
#endif
constexpr bool kDfStagger = OKKT_DF_STAGGER != 0;   // waves 4 - 7 of a worker request the next operand chunk half a chunk behind waves 0 - 3
// The experiments of round 6 -- the block row in lockstep with its diagonal block (OKKT_DF_PROG), macro tiles (OKKT_DF_MACRO), chained update tasks
// (OKKT_DF_CHAIN_BUILD) -- are bitwise equal, not faster, and NOT part of the product library: a persistent kernel pays for every role it carries
// (S-C3 / S-C5 1 % slower with the first two compiled in and switched off, S-metric 2.5 % with the third).  `make` builds them into a second
// library, libonephase_kkt_exp.so (all three on), which the GPU tests of those variants load through OKKT_LIB_PATH.
#ifndef OKKT_DF_PROG
#define OKKT_DF_PROG 0
#endif
constexpr bool kDfProg = OKKT_DF_PROG != 0;   // round 6: the diagonal block reports its finished 32-column blocks (front_device.h, diag2_body<LPROG>) and TU follows in lockstep (df_tu_lock)
#ifndef OKKT_DF_MACRO
#define OKKT_DF_MACRO 0
#endif
#ifndef OKKT_DF_ROT
#define OKKT_DF_ROT 0
#endif
#ifndef OKKT_DF_CHAIN_BUILD
#define OKKT_DF_CHAIN_BUILD 0
#endif
// chained update tasks (df_syrk_chain) and the half-work timing experiment are compiled in only on request: with the role in the kernel and switched
// OFF at run time every other role got slower (S-metric 16.97 -> 17.39 ms, S-C3 + 1.5 %, S-C5 + 1.3 % on one box: twice the scalar-register spills,
// a longer instruction stream for the persistent workers) -- round 6, profiles/r06_chain_role_in_kernel_ab.txt
constexpr bool kDfChain = OKKT_DF_CHAIN_BUILD != 0;
#ifndef OKKT_DF_LOG_BUILD
#define OKKT_DF_LOG_BUILD 0
#endif
// the per-task time stamps (OKKT_DEBUG_DATAFLOW=16 + OKKT_DF_LOG) and the switches that skip task bodies (1 / 2 / 4) are instrumentation of the
// same kind: with them compiled out S-metric is 1.6 % faster (17.06 against 17.33 ms), S-C3 0.8 %, S-C5 1.6 % (82 instead of 101 scalar-register
// spills, a shorter instruction stream).  libonephase_kkt_log.so (the product kernel + this instrumentation) and the experiments library have them;
// OKKT_DEBUG_DATAFLOW=8 (synchronise and report every launch) works everywhere.
constexpr bool kDfLog = OKKT_DF_LOG_BUILD != 0;
#ifndef OKKT_DF_PIPE
#define OKKT_DF_PIPE 0
#endif
#ifndef OKKT_DF_MULTI
#define OKKT_DF_MULTI 1
#endif
// update tasks that carry several row tiles (OKKT_DF_ROWS, OKKT_DF_ROWS_BIG: measured slower since round 4) need a second set of C-tile registers in the
// update role.  -DOKKT_DF_MULTI=0 leaves it out; measured (round 6, four A/B pairs on one box): the kernel WITH it is the faster one -- S-metric 17.07 against
// 17.17 ms, S-C3 2.95 against 2.99 -- register allocation, not logic: kept
constexpr bool kDfMulti = OKKT_DF_MULTI != 0;
constexpr bool kDfPipe = OKKT_DF_PIPE != 0;      // update tasks: operand fragments requested one step ahead of their MFMAs, reads and waits by hand (df_syrk_tiles); bitwise equal, no gain in the kernel: off
constexpr bool kDfRot = OKKT_DF_ROT != 0;   // update tasks: the column fragments of a k-step from ONE LDS read + lane rotations (df_syrk_tiles); bitwise equal, 12 % SLOWER (see there)
constexpr bool kDfMacro = OKKT_DF_MACRO != 0;   // update tasks on pairs of row tiles as one macro tile (df_syrk_macro)
constexpr int kDfDiagMfmaWaves = 6;   // MFMA waves of the diagonal-block factorisation in a worker (four or six: the same time)
constexpr int kDfKC = OKKT_DF_KC;           // panel columns per ring slot of the update tasks
constexpr int kDfStages = OKKT_DF_STAGES;   // operand ring of the update tasks: 16-column chunks in LDS (one workgroup per CU: nobody else covers a chunk that is late)

// 16-byte write-through store (global_store_dwordx4 ... sc1): the C tiles leave the CU at the rate of plain stores and are
// visible to every other CU once the storing wave's vmcnt has drained
__device__ __forceinline__ void st_sc1_f64x2(double* p, d2_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ int ld_state(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// first row / column of block b of a front with KB pivot blocks: 128 b inside the pivot block, k + 128 (b - KB) behind it
__device__ __forceinline__ int df_block_lo(int b, int KB, int k, int f) { return b < KB ? b * 128 : min(k + (b - KB) * 128, f); }

// ---- T: rows [r0, rlim) (at most 128) of the panel below diagonal block q ---------------------------------------------------
// k_big_trsm's body (front_device.h: trsm_body) for a workgroup of eight waves, 16 rows each; every load behind the task's
// acquire is a plain load, W and L leave with sc1 stores.
__device__ __forceinline__ void df_trsm_tile(const DevPlan& P, int s, int q, int r0, int rlim, double* sm) {
  constexpr int NBLK = 4, NB = 128, NPAIR = NBLK * (NBLK + 1) / 2;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // opaque: nothing derived from the lane id is hoisted out of the worker's loop and kept live across the other roles
  const int tid = tid_, lane = tid & 63, wv = tid >> 6;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = q * NB;
  const int nb = min(NB, k - j0);
  double* F = P.arena + P.front_pos[s];
  double* Wb = P.wbuf + P.wbuf_pos[s] + (size_t)j0 * f;
  const double* X = P.invl + P.invl_pos[s] + (size_t)q * NB * NB;
  double* rdv = sm + NPAIR * kIB * kIB;
  const int row = r0 + wv * 16 + (lane & 15);
  const int rowc = min(row, f - 1);
  const bool valid = row < rlim;
  const int lk = lane >> 4, li = lane & 3;
  double t[NBLK * 8];
#pragma unroll
  for (int qq = 0; qq < NBLK * 8; ++qq) {
    const int c = 4 * qq + lk;
    t[qq] = keep_f64(F[(size_t)(j0 + min(c, nb - 1)) * f + rowc], c < nb && valid);
  }
  {
    const int e = tid * 2;                 // 2 consecutive rows of one column per thread and block
    const int cc = e / kIB, rr = e - cc * kIB;
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
      for (int bp = 0; bp <= bi; ++bp) {
        double v[2];
        const int gr = bi * kIB + rr, gc = bp * kIB + cc;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const double* src = bp == bi ? X + (gr + u) + (size_t)gc * NB
                                       : F + (size_t)(j0 + min(gc, nb - 1)) * f + j0 + min(gr + u, nb - 1);
          v[u] = keep_f64(*src, gr + u < nb && gc < nb);
        }
        double* dst = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB + e;
        dst[0] = v[0]; dst[1] = v[1];
      }
    if (tid < NB) rdv[tid] = tid < nb ? 1.0 / P.dvals[col0 + j0 + tid] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int bi = 0; bi < NBLK; ++bi) {
#pragma unroll
    for (int bp = 0; bp < bi; ++bp) {
      const double* Lb = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB;
#pragma unroll
      for (int gp = 0; gp < 8; ++gp)
#pragma unroll
        for (int g = 0; g < 8; ++g)
          t[bi * 8 + gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Lb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bp * 8 + g], t[bi * 8 + gp], 0, 0, 1 /* neg A */);
    }
    const double* Xb = sm + (bi * (bi + 1) / 2 + bi) * kIB * kIB;
    double wt[8];
#pragma unroll
    for (int gp = 0; gp < 8; ++gp) {
      wt[gp] = 0.0;
#pragma unroll
      for (int g = 0; g <= gp; ++g)   // X_ii is lower triangular
        wt[gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Xb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bi * 8 + g], wt[gp], 0, 0, 0);
    }
#pragma unroll
    for (int gp = 0; gp < 8; ++gp) t[bi * 8 + gp] = wt[gp];
  }
  if (valid) {
#pragma unroll
    for (int qq = 0; qq < NBLK * 8; ++qq) {
      const int c = 4 * qq + lk;
      if (c < nb) {
        st_agent_f64(&Wb[(size_t)c * f + row], t[qq]);
        st_agent_f64(&F[(size_t)(j0 + c) * f + row], t[qq] * rdv[c]);
      }
    }
  }
}

// a wait inside a task (the diagonal block a TL task solves against): the worker's protocol -- wave 0 polls (bounded), acquires, the
// workgroup meets.  false: the launch is stopping (delta loop) or a wait has run into its bound
__device__ __forceinline__ bool df_await(const DevPlan& P, const int* state, int least, int* flag) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave == 0) {
    int ok = 1, spins = 0;
    long long t0w = 0;
    for (;;) {
      if (__builtin_amdgcn_readfirstlane(ld_state(state)) >= least) break;
      const int stop = P.want_neg >= 0 ? __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
      const int dead = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (stop | dead) { ok = 0; break; }
      if (wait_expired(spins, t0w)) {
        if ((threadIdx.x & 63) == 0) atomicExch(&P.counters[5], 1ull);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *flag = ok;
  }
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*flag) != 0;
}

// ---- TL: the last update of tile (i, q) (panel q - 1) and T(i, q) in one task, q >= 1 ------------------------------------------------
// A block row below the chain's advances one block column per chain step: T(i, q - 1) -> U(i, q; panel q - 1) -> T(i, q), two tasks and two
// hand-offs (30 + 27 + 2 x 3 us) per 63-us step -- every row was as slow as the chain itself, any queueing made it slower, and the chain
// waits for the rows that enter its cone: the root of the metric workload ran at 87 - 94 us per block column while updates were
// plentiful.  Here the update runs in the panel solve's own register layout (a wave owns 16 rows x 128 columns: the 16 x 4 strips are
// the accumulators, W(i, q - 1) in the same layout is the B operand, L(q, q - 1) staged in LDS the A operand: 1024 MFMAs per wave) and
// BEFORE D(q) has arrived -- the task is popped when its tile has received panel q - 2 and block rows i and q of panel q - 1 are there,
// waits for D(q) inside like TU, stages the diagonal block's pieces over the L image and solves.  The products reach every entry in
// ascending panel order, k ascending within the panel, as in the update task: bitwise the same tile, then df_trsm_tile's solve.
// W and L leave through LDS as 16-byte write-through stores (the 8-byte ones of df_trsm_tile cost 9 us per tile).
constexpr int kTlLd = 132;           // leading dimension of the L(q, q - 1) image: the four k-slices of an A fragment on disjoint banks
constexpr size_t kDfTlLds = std::max((size_t)128 * kTlLd * sizeof(double), (size_t)(128 * kSyrkLd + 128) * sizeof(double));
__device__ __forceinline__ bool df_tl_tile(const DevPlan& P, int s, int q, int r0, int rlim, const int* dstate, int* s_flag, double* sm, long long* marks) {
  constexpr int NBLK = 4, NB = 128, NPAIR = NBLK * (NBLK + 1) / 2;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wv = tid >> 6;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = q * NB;
  const int nb = min(NB, k - j0);
  double* F = P.arena + P.front_pos[s];
  double* Wb = P.wbuf + P.wbuf_pos[s] + (size_t)j0 * f;
  const double* Wprev = Wb - (size_t)NB * f;              // W of panel q - 1 (a whole block column)
  const double* X = P.invl + P.invl_pos[s] + (size_t)q * NB * NB;
  const int trow = wv * 16 + (lane & 15);
  const int row = r0 + trow;
  const int rowc = min(row, f - 1);
  const bool valid = row < rlim;
  const int lk = lane >> 4, li = lane & 3;
  double t[NBLK * 8], w[NBLK * 8];
#pragma unroll
  for (int qq = 0; qq < NBLK * 8; ++qq) {
    const int c = 4 * qq + lk;
    t[qq] = keep_f64(F[(size_t)(j0 + min(c, nb - 1)) * f + rowc], c < nb && valid);
    w[qq] = keep_f64(Wprev[(size_t)c * f + rowc], valid);
  }
  // L(q, q - 1): rows [j0, j0 + nb) of panel q - 1, entry (c, kk) at c + kk * kTlLd
  {
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int idx = it * kDfThreadsC + tid;
      const int kk = idx >> 6, c2 = (idx & 63) * 2;
      const double* src = F + (size_t)(j0 - NB + kk) * f + j0;
      d2_t v;
      v[0] = keep_f64(src[min(c2, nb - 1)], c2 < nb);
      v[1] = keep_f64(src[min(c2 + 1, nb - 1)], c2 + 1 < nb);
      __builtin_memcpy(sm + (size_t)kk * kTlLd + c2, &v, 16);
    }
  }
  __syncthreads();
  if (marks && tid == 0) marks[0] = wall_clock64();        // operands in registers and LDS
  // t(rows, c) -= sum_kk W(rows, kk) L(c, kk): k-strips ascending, the order of the update task's k-steps
  // (fully unrolled: a rolled loop over g indexes w[] dynamically, which puts the array into scratch memory -- a scratch load and a
  // vmcnt(0) per 64 MFMAs)
#pragma unroll
  for (int g = 0; g < NBLK * 8; ++g) {
#pragma unroll
    for (int gp = 0; gp < NBLK * 8; ++gp)
      t[gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(sm[(gp * 4 + li) + (size_t)(g * 4 + lk) * kTlLd], w[g], t[gp], 0, 0, 1 /* neg A */);
  }
  if (marks && tid == 0) marks[1] = wall_clock64();        // tile updated
  if (!df_await(P, dstate, q + 1, s_flag)) return false;   // D(q); the barrier inside: every wave is done with the L image
  if (marks && tid == 0) marks[2] = wall_clock64();        // D(q) has arrived
  double* rdv = sm + NPAIR * kIB * kIB;
  {
    const int e = tid * 2;
    const int cc = e / kIB, rr = e - cc * kIB;
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
      for (int bp = 0; bp <= bi; ++bp) {
        double v[2];
        const int gr = bi * kIB + rr, gc = bp * kIB + cc;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const double* src = bp == bi ? X + (gr + u) + (size_t)gc * NB
                                       : F + (size_t)(j0 + min(gc, nb - 1)) * f + j0 + min(gr + u, nb - 1);
          v[u] = keep_f64(*src, gr + u < nb && gc < nb);
        }
        double* dst = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB + e;
        dst[0] = v[0]; dst[1] = v[1];
      }
    if (tid < NB) rdv[tid] = tid < nb ? 1.0 / P.dvals[col0 + j0 + tid] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int bi = 0; bi < NBLK; ++bi) {
#pragma unroll
    for (int bp = 0; bp < bi; ++bp) {
      const double* Lb = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB;
#pragma unroll
      for (int gp = 0; gp < 8; ++gp)
#pragma unroll
        for (int g = 0; g < 8; ++g)
          t[bi * 8 + gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Lb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bp * 8 + g], t[bi * 8 + gp], 0, 0, 1 /* neg A */);
    }
    const double* Xb = sm + (bi * (bi + 1) / 2 + bi) * kIB * kIB;
    double wt[8];
#pragma unroll
    for (int gp = 0; gp < 8; ++gp) {
      wt[gp] = 0.0;
#pragma unroll
      for (int g = 0; g <= gp; ++g)   // X_ii is lower triangular
        wt[gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Xb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bi * 8 + g], wt[gp], 0, 0, 0);
    }
#pragma unroll
    for (int gp = 0; gp < 8; ++gp) t[bi * 8 + gp] = wt[gp];
  }
  const double myrd = tid < NB ? rdv[tid] : 0.0;
  __syncthreads();                       // every wave is done with the staged blocks and the reciprocals
  double* Wl = sm;                       // [128 panel columns][kSyrkLd]: W(r0 + r, j0 + p) at p * kSyrkLd + r
  double* rd2 = sm + (size_t)128 * kSyrkLd;
#pragma unroll
  for (int qq = 0; qq < NBLK * 8; ++qq) Wl[(size_t)(4 * qq + lk) * kSyrkLd + trow] = t[qq];      // rows past the block are zero
  if (tid < NB) rd2[tid] = myrd;
  __syncthreads();
  if (marks && tid == 0) marks[3] = wall_clock64();        // solved, W in LDS
  {
    const int nrow = min(rlim - r0, 128);
    for (int idx = tid; idx < (nb << 6); idx += kDfThreadsC) {
      const int p = idx >> 6, x2 = (idx & 63) * 2;
      d2_t wv2;
      __builtin_memcpy(&wv2, Wl + (size_t)p * kSyrkLd + x2, 16);
      const double rp = rd2[p];
      const d2_t l = (d2_t){wv2[0] * rp, wv2[1] * rp};
      double* wdst = Wb + (size_t)p * f + r0 + x2;
      double* ldst = F + (size_t)(j0 + p) * f + r0 + x2;
      if (x2 + 1 < nrow) { st_sc1_f64x2(wdst, wv2); st_sc1_f64x2(ldst, l); }
      else if (x2 < nrow) { st_agent_f64(wdst, wv2[0]); st_agent_f64(ldst, l[0]); }
    }
  }
  return true;
}

// ---- TU: T(q + 1, q) and the update of the diagonal tile (q + 1, q + 1) by panel q in one task -------------------------------
// The two steps between the diagonal blocks of consecutive block columns.  The task is popped beside D(q): the rows of tile
// (q + 1, q) and the diagonal tile are in flight while D(q) runs; then the task waits for D(q) (second wait, same protocol as the
// worker's), solves its 128 rows like df_trsm_tile, keeps W in LDS (128 x 128, the layout of the trailing update's operand
// slots; L = W * (1 / d) is formed on the fly, the very product that is stored) and applies it to the diagonal tile with the
// trailing update's MFMA loop.  Same operations per entry in the same order as T followed by U: bitwise the same numbers.
//
// Both steps run at the FP64 matrix rate of ONE CU (7 us for the rows, 9 - 11 us for the ten 32 x 32 sub-tiles of the lower
// triangle), so the block row may be split between two workers (`part`):
//   part 1 (task TA): rows [r0, r0 + 64) -- solves them, stores their W and L (half state 1), updates the sub-tiles that need
//                     nothing else ((0,0), (1,0), (1,1) of the 4 x 4 grid), stores them (half state 2: published by the worker);
//   part 2 (task TU): rows [r0 + 64, rlim) -- solves them, fetches the W of the upper rows once half state 1 is seen, updates the
//                     seven sub-tiles of its rows, takes the three of part 1 from memory (half state 2) and goes on as the
//                     unsplit task does (the diagonal block's factorisation from LDS, or the tile's publication);
//   part 0: everything in one task (block rows of at most 64 rows, OKKT_DF_SPLIT_TU=0).
// The half state lives in the unused upper slot (q, q + 1) of the front's tile states.
// units (row block | column half-block << 2 of the 4 x 8 grid of 32 x 16 units; 255 = none) per part, wave and slot: see df_tu_tile
__device__ const unsigned char kTuUnits[3][8][3] = {
    {{0, 10, 19}, {4, 14, 23}, {1, 18, 27}, {5, 22, 31}, {9, 3, 255}, {13, 7, 255}, {2, 11, 255}, {6, 15, 255}},
    {{0, 255, 255}, {4, 255, 255}, {1, 255, 255}, {5, 255, 255}, {9, 255, 255}, {13, 255, 255}, {255, 255, 255}, {255, 255, 255}},
    {{18, 2, 255}, {22, 6, 255}, {19, 10, 255}, {23, 14, 255}, {27, 3, 255}, {31, 7, 255}, {255, 11, 255}, {255, 15, 255}}};
constexpr size_t kDfTuLds = ((size_t)128 * kSyrkLd + 128) * sizeof(double);
constexpr int kDfTileLd = 130;       // leading dimension of the diagonal tile handed from TU to D through LDS (even: 16-byte rows pairs stay aligned)
__device__ __forceinline__ bool df_tu_tile(const DevPlan& P, int s, int q, int r0, int rlim, const int* dstate, int dval, int* diag_state, int* row_state, int* half_state, int part,
                                           bool with_d, int* s_flag, double* sm, long long* marks) {
  constexpr int NBLK = 4, NB = 128, NPAIR = NBLK * (NBLK + 1) / 2;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wv = tid >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(wv);
  const int col0 = P.sn_col0[s];
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = q * NB;
  // q + 1 < KB: panel q is a whole block column (nb = 128)
  double* F = P.arena + P.front_pos[s];
  double* Wb = P.wbuf + P.wbuf_pos[s] + (size_t)j0 * f;
  const double* X = P.invl + P.invl_pos[s] + (size_t)q * NB * NB;
  double* rdv = sm + NPAIR * kIB * kIB;
  // the rows this task solves: sixteen per wave, eight waves for a whole block row, four for a half
  const int roff = part == 2 ? 64 : 0;
  const int hrows = part == 0 ? 128 : 64;
  const bool solver = part == 0 || wave < 4;
  const int trow = roff + (part == 0 ? wv : (wv & 3)) * 16 + (lane & 15);      // row of the tile
  const int row = r0 + trow;
  const int rowc = min(row, f - 1);
  const bool valid = solver && row < rlim;
  const int lk = lane >> 4, li = lane & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  double t[NBLK * 8];
#pragma unroll
  for (int qq = 0; qq < NBLK * 8; ++qq) t[qq] = 0.0;
  if (solver) {
#pragma unroll
    for (int qq = 0; qq < NBLK * 8; ++qq) {
      const int c = 4 * qq + lk;
      t[qq] = keep_f64(F[(size_t)(j0 + c) * f + rowc], valid);
    }
  }
  // The diagonal tile (rows and columns [r0, rlim)) is updated in UNITS of 32 rows x 16 columns -- 8 MFMAs per k-step of four
  // panel columns, 4096 cycles of the SIMD's matrix pipe (1.75 us).  Only the lower triangle counts: 20 units of 32, dealt over the
  // waves so that every SIMD (waves w and w + 4 share one: scripts/simd_probe.hip) has TWO waves issuing -- one wave alone gets an
  // FP64 MFMA through every 32 cycles instead of every 16 (measured here: 512 MFMAs of one wave 6.9 us; the 64 x 32 pieces of the
  // first version, one or two per SIMD, took 10.7 us for the tile):
  //   part 0: 3, 3, 3, 3, 2, 2, 2, 2 units per wave (five per SIMD);
  //   part 1: its six units (row blocks 0 and 1) on waves 0-5;
  //   part 2: the six units of its own rows and columns on waves 0-5 (slot 0: they run while the workgroup waits for the other
  //           half's W), the eight left of them on waves 0-7 (slot 1).
  constexpr int NU = 3;
  int ucode[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) ucode[u] = __builtin_amdgcn_readfirstlane((int)kTuUnits[part][wave][u]);
  auto unit_on = [&](int u) { return ucode[u] != 255 && r0 + 32 * (ucode[u] & 3) < rlim && r0 + 16 * (ucode[u] >> 2) < rlim; };
  double acc[NU][4][2];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) { acc[u][cg][0] = 0.0; acc[u][cg][1] = 0.0; }
    if (!unit_on(u)) continue;
    const int r = r0 + 32 * (ucode[u] & 3) + 2 * l15;
    const int rcl = min(r, f - 2);
    const int shift = r - rcl;
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      const int c = r0 + 16 * (ucode[u] >> 2) + cg * 4 + l4;
      const double* colp = F + (size_t)min(c, rlim - 1) * f;      // (columns of a pivot tile: clamped inside the panel)
      d2_t v;
      __builtin_memcpy(&v, colp + rcl, 16);
      const double e0 = shift == 0 ? v[0] : v[1];
      acc[u][cg][0] = keep_f64(e0, r < rlim && c < rlim && r >= c);
      acc[u][cg][1] = keep_f64(v[1], shift == 0 && r + 1 < rlim && c < rlim && r + 1 >= c);
    }
  }
  // a wait inside the task (the diagonal block of panel q; the other half's states): the worker's protocol, one flag word per wait
  auto await = [&](const int* state, int least, int* flag, int poller) -> bool {
    if (wave == poller) {
      int ok = 1, spins = 0;
      long long t0w = 0;
      for (;;) {
        if (__builtin_amdgcn_readfirstlane(ld_state(state)) >= least) break;
        const int stop = P.want_neg >= 0 ? __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
        const int dead = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (stop | dead) { ok = 0; break; }
        if (wait_expired(spins, t0w)) {
          if (lane == 0) atomicExch(&P.counters[5], 1ull);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *flag = ok;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*flag) != 0;
  };
  if (!await(dstate, dval, s_flag, 0)) return false;
  if (marks && tid == 0) marks[0] = wall_clock64();        // D(q) has arrived
  {
    const int e = tid * 2;
    const int cc = e / kIB, rr = e - cc * kIB;
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
      for (int bp = 0; bp <= bi; ++bp) {
        double v[2];
        const int gr = bi * kIB + rr, gc = bp * kIB + cc;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const double* src = bp == bi ? X + (gr + u) + (size_t)gc * NB : F + (size_t)(j0 + gc) * f + j0 + gr + u;
          v[u] = *src;
        }
        double* dst = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB + e;
        dst[0] = v[0]; dst[1] = v[1];
      }
    if (tid < NB) rdv[tid] = 1.0 / P.dvals[col0 + j0 + tid];
  }
  __syncthreads();
#ifdef OKKT_TU_STAGE_MARK
  if (marks && tid == 0) marks[1] = wall_clock64();        // (variant build) the staged blocks are in LDS
#define OKKT_TU_SKIP_MARK1 1
#endif
  if (solver) {
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi) {
#pragma unroll
      for (int bp = 0; bp < bi; ++bp) {
        const double* Lb = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB;
#pragma unroll
        for (int gp = 0; gp < 8; ++gp)
#pragma unroll
          for (int g = 0; g < 8; ++g)
            t[bi * 8 + gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Lb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bp * 8 + g], t[bi * 8 + gp], 0, 0, 1 /* neg A */);
      }
      const double* Xb = sm + (bi * (bi + 1) / 2 + bi) * kIB * kIB;
      double wt[8];
#pragma unroll
      for (int gp = 0; gp < 8; ++gp) {
        wt[gp] = 0.0;
#pragma unroll
        for (int g = 0; g <= gp; ++g)
          wt[gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Xb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bi * 8 + g], wt[gp], 0, 0, 0);
      }
#pragma unroll
      for (int gp = 0; gp < 8; ++gp) t[bi * 8 + gp] = wt[gp];
    }
  }
#ifndef OKKT_TU_SKIP_MARK1
  if (marks && tid == 0) marks[1] = wall_clock64();        // rows solved
#endif
  const double myrd = tid < NB ? rdv[tid] : 0.0;
  __syncthreads();                       // every wave is done with the staged blocks and the reciprocals
  double* Wl = sm;                       // [128 panel columns][kSyrkLd]: W(r0 + r, j0 + p) at p * kSyrkLd + r
  double* rd2 = sm + (size_t)128 * kSyrkLd;
  if (solver) {
#pragma unroll
    for (int qq = 0; qq < NBLK * 8; ++qq) Wl[(size_t)(4 * qq + lk) * kSyrkLd + trow] = t[qq];      // rows past the block are zero
  }
  if (tid < NB) rd2[tid] = myrd;
  __syncthreads();
  if (marks && tid == 0 && part == 0) marks[2] = wall_clock64();        // W in LDS (a split task: the hand-over of the upper rows' W, below)
  // W and L = W D^-1 of the task's rows leave from LDS: 16-byte write-through stores of two consecutive rows, 1 KiB (512 B for
  // a half) of one column per wave instruction (the 8-byte sc1 stores straight from the solve's register layout took 9 us of the 42
  // between two diagonal blocks).
  auto store_wl = [&](int first_thread, int nthreads) {
    const int nrow = min(rlim - r0, roff + hrows);
    const int sh = part == 0 ? 6 : 5;                      // log2 of the row pairs per column
    if (tid < first_thread || tid >= first_thread + nthreads) return;
    for (int idx = tid - first_thread; idx < (128 << sh); idx += nthreads) {
      const int p = idx >> sh, x2 = roff + (idx & ((1 << sh) - 1)) * 2;
      d2_t w;
      __builtin_memcpy(&w, Wl + (size_t)p * kSyrkLd + x2, 16);
      const double rp = rd2[p];
      const d2_t l = (d2_t){w[0] * rp, w[1] * rp};
      double* wdst = Wb + (size_t)p * f + r0 + x2;
      double* ldst = F + (size_t)(j0 + p) * f + r0 + x2;
      if (x2 + 1 < nrow) { st_sc1_f64x2(wdst, w); st_sc1_f64x2(ldst, l); }
      else if (x2 < nrow) { st_agent_f64(wdst, w[0]); st_agent_f64(ldst, l[0]); }
    }
  };
  // the update of unit u: the operand fragments of the next k-step are requested before the products of the current one are issued
  auto update_unit = [&](auto uc) {
    constexpr int u = decltype(uc)::value;
    if (!unit_on(u)) return;
    const double* bw = Wl + 32 * (ucode[u] & 3) + 2 * l15;
    const double* bl = Wl + 16 * (ucode[u] >> 2) + (lane & 3);
    double nrd, nav[4];
    d2_t nbv;
    auto fetch = [&](int kk) {
      nrd = rd2[kk * 4 + l4];
      __builtin_memcpy(&nbv, bw + (kk * 4 + l4) * kSyrkLd, 16);
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) nav[cg] = bl[(kk * 4 + l4) * kSyrkLd + cg * 4];
    };
    fetch(0);
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; ++kk) {
      double av[4];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const d2_t bv = nbv;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) av[cg] = nav[cg] * nrd;
      fetch(min(kk + 1, NB / 4 - 1));
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        acc[u][cg][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[cg], bv[0], acc[u][cg][0], 0, 0, 1 /* neg A */);
        acc[u][cg][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[cg], bv[1], acc[u][cg][1], 0, 0, 1 /* neg A */);
      }
    }
  };
  if (part == 1) {
    // the other half waits for these rows: out, drained, published
    store_wl(0, kDfThreadsC);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) __hip_atomic_store(half_state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (marks && tid == 0) marks[2] = wall_clock64();      // W and L of the upper rows stored, drained, published
    update_unit(std::integral_constant<int, 0>());
  } else if (part == 2) {
    // W and L of these rows to memory (every wave its share: one wave alone has 64 stores in flight at most and took 8 us over
    // them); then waves 0-5 the units of its own rows and columns, wave 7 the wait for the other half's W
    store_wl(6 * 64, 128);
    update_unit(std::integral_constant<int, 0>());
    if (!await(half_state, 1, s_flag + 1, 0)) return false;
    if (marks && tid == 0) marks[2] = wall_clock64();      // ... and seen by the other half
    // W of the upper 64 rows (the column operand of this half's sub-tiles left of the diagonal) from memory into the image
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int idx = it * kDfThreadsC + tid;
      const int p = idx >> 5, x2 = (idx & 31) * 2;
      d2_t w;
      __builtin_memcpy(&w, Wb + (size_t)p * f + r0 + x2, 16);
      __builtin_memcpy(Wl + (size_t)p * kSyrkLd + x2, &w, 16);
    }
    __syncthreads();
    update_unit(std::integral_constant<int, 1>());
  } else {
    update_unit(std::integral_constant<int, 0>());
    update_unit(std::integral_constant<int, 1>());
    update_unit(std::integral_constant<int, 2>());
  }
  // the wave's units to memory (write-through) ...
  auto store_piece = [&]() {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (!unit_on(u)) continue;
      const int r = r0 + 32 * (ucode[u] & 3) + 2 * l15;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        const int c = r0 + 16 * (ucode[u] >> 2) + cg * 4 + l4;
        if (c >= rlim) continue;
        double* colp = F + (size_t)c * f;
        if (r + 1 < rlim && r >= c) {
          st_sc1_f64x2(colp + r, (d2_t){acc[u][cg][0], acc[u][cg][1]});
        } else {
          if (r < rlim && r >= c) st_agent_f64(colp + r, acc[u][cg][0]);
          if (r + 1 < rlim && r + 1 >= c) st_agent_f64(colp + r + 1, acc[u][cg][1]);
        }
      }
    }
  };
  // ... or into the LDS tile the diagonal-block factorisation reads (column-major, leading dimension kDfTileLd, over the W image)
  auto piece_to_lds = [&]() {
    double* Tl = sm;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (!unit_on(u)) continue;
      const int r = 32 * (ucode[u] & 3) + 2 * l15;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        const int c = 16 * (ucode[u] >> 2) + cg * 4 + l4;
        const d2_t v = (d2_t){acc[u][cg][0], acc[u][cg][1]};
        __builtin_memcpy(Tl + (size_t)c * kDfTileLd + r, &v, 16);
      }
    }
  };
  if (part == 1) {
    // sub-tiles (0,0), (1,0), (1,1): to memory; the worker drains, meets and publishes half state 2
    if (marks && tid == 0) marks[3] = wall_clock64();
    store_piece();
    return true;
  }
  if (!with_d) {
    // the diagonal tile first: it is what D(q + 1) waits for.  Every storing wave drains, the workgroup meets, one wave publishes.
    store_piece();
    if (part == 2 && !await(half_state, 2, s_flag + 2, 0)) return false;      // ... the other half's sub-tiles are part of the tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) __hip_atomic_store(diag_state, dval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // tile (q + 1, q + 1) has received panel q (= q + 1 tasks)
    if (marks && tid == 0) marks[3] = wall_clock64();      // diagonal tile updated and published
    if (part == 0) store_wl(0, kDfThreadsC);
  } else {
    // TU(q) + D(q + 1) in one task: the block row's W and L go out (the other rows' updates of block column q + 1 wait for them:
    // stored ahead of the tile update they delayed it by 10 us, published from inside the factorisation they came 4 us late --
    // both measured slower on S-C3), then the updated diagonal tile is handed to the diagonal-block factorisation through LDS
    // -- no trip through HBM, no hand-off
    if (marks && tid == 0) marks[3] = wall_clock64();      // diagonal tile updated
    if (part == 0) store_wl(0, kDfThreadsC);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                     // every wave has read its part of the W image
    piece_to_lds();
    if (part == 2) {
      if (!await(half_state, 2, s_flag + 2, 0)) return false;
      // the other half's sub-tiles: rows and columns [0, 64) of the tile, from memory
      double* Tl = sm;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * kDfThreadsC + tid;
        const int c = idx >> 5, x2 = (idx & 31) * 2;
        d2_t v;
        __builtin_memcpy(&v, F + (size_t)(r0 + c) * f + r0 + x2, 16);
        v[0] = keep_f64(v[0], x2 >= c);
        v[1] = keep_f64(v[1], x2 + 1 >= c);
        __builtin_memcpy(Tl + (size_t)c * kDfTileLd + x2, &v, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) __hip_atomic_store(row_state, dval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // tile (q + 1, q): W and L of the block row are in memory
  }
  return true;
}

// ---- TU in lockstep with D(q) (round 6) ---------------------------------------------------------------------------------------------
// The chain between two diagonal blocks was D(q) 32 us -> TU(q) 28 us (stage L(q, q) and the 32 x 32 inverses, solve the 128 rows, update
// the diagonal tile (q + 1, q + 1)) -> D(q + 1): 60 us per 128 pivot columns whatever is left to update -- the bound of every front's tail and
// of every configuration but the metric one.  Nothing in TU needs ALL of D(q): the blocked substitution W_b = (A_b - sum_{p < b} W_p L_bp^T)
// X_bb^T needs column block p of L(q, q) when it reaches block p, and the tile update is a sum over the panel's columns in ascending
// order.  diag2_body<LPROG> reports its 32-column blocks as their L entries and pivots reach memory (sub-state b + 1 in the slot
// (q, q + 1) of the tile states); this task, popped beside D(q) as before, follows one block behind.  Per column block b:
//   1. the report arrives -> every thread stages L(b' > b, b) and, for the inverting wave, the diagonal block L(b, b);
//   2. wave 7 inverts L(b, b) -- the column-oriented forward substitution of diag2_body's tail, the same operations in the same order:
//      bitwise the X_bb that D(q) itself computes later for everybody else -- WHILE waves 0 - 6 store W and L of block b - 1 and apply
//      W_(b - 1) to the diagonal tile (8 k-steps of the units' MFMA loop, k ascending as before: bitwise the same tile);
//   3. W_b = t_b X_bb^T, W_b into the LDS image (two images: the next block's step 2 reads this one), then RIGHT-looking t_b' -= W_b L_b'b^T
//      for the later blocks (every entry receives its terms in the order of the left-looking df_tu_tile: bitwise the same W).
// Behind D(q)'s last report only block 3 is left.  One worker per block row: the TA / TU split bought 6 us of a 28-us step.
// The inverses cost this worker 3.2 us per block on a wave that does nothing else; inside D(q) they cost every micro-step (see
// front_device.h).  The twenty 32 x 16 units of the tile are dealt to waves 0 - 6 (kTuLockUnits).
__device__ const unsigned char kTuLockUnits[8][3] = {{0, 10, 19}, {4, 14, 23}, {1, 18, 27}, {5, 22, 31}, {9, 3, 2}, {13, 7, 6}, {11, 15, 255}, {255, 255, 255}};
constexpr size_t kDfTuLockLds = ((size_t)3 * kIB * kIB + 2 * 32 * kXld + 64 + 2 * ((size_t)32 * kSyrkLd + 32)) * sizeof(double);
__device__ __forceinline__ bool df_tu_lock(const DevPlan& P, int s, int q, int r0, int rlim, const int* sub_state, int dval, int* diag_state, int* row_state,
                                           bool with_d, int* s_flag, double* sm, long long* marks) {
  constexpr int NBLK = 4, NB = 128;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wv = tid >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(wv);
  const int col0 = P.sn_col0[s];
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = q * NB;      // q + 1 < KB: panel q is a whole block column
  double* F = P.arena + P.front_pos[s];
  double* Wb = P.wbuf + P.wbuf_pos[s] + (size_t)j0 * f;
  double* Sb = sm;                                    // three staged 32 x 32 blocks: L(b', b) for b' = b + 1 .. 3 (leading dimension kIB)
  double* Li = sm + 3 * kIB * kIB;                    // L(b, b) for the inverting wave, leading dimension kXld
  double* Xi = Li + 32 * kXld;                        // its inverse, leading dimension kXld
  double* rdv = Xi + 32 * kXld;                       // 2 x 32 reciprocal pivots (block parity)
  double* Wl0 = rdv + 64;                             // two W images [32 panel columns][kSyrkLd] (block parity): W(r0 + r, j0 + 32 b + p) at p * kSyrkLd + r
  constexpr size_t kWlStride = (size_t)32 * kSyrkLd + 32;
  const int trow = wv * 16 + (lane & 15);             // row of the tile this lane solves
  const int row = r0 + trow;
  const int rowc = min(row, f - 1);
  const bool valid = row < rlim;
  const int lk = lane >> 4, li = lane & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  double t[NBLK * 8];
#pragma unroll
  for (int qq = 0; qq < NBLK * 8; ++qq) {
    const int c = 4 * qq + lk;
    t[qq] = keep_f64(F[(size_t)(j0 + c) * f + rowc], valid);
  }
  constexpr int NU = 3;
  int ucode[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) ucode[u] = __builtin_amdgcn_readfirstlane((int)kTuLockUnits[wave][u]);
  auto unit_on = [&](int u) { return ucode[u] != 255 && r0 + 32 * (ucode[u] & 3) < rlim && r0 + 16 * (ucode[u] >> 2) < rlim; };
  double acc[NU][4][2];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) { acc[u][cg][0] = 0.0; acc[u][cg][1] = 0.0; }
    if (!unit_on(u)) continue;
    const int r = r0 + 32 * (ucode[u] & 3) + 2 * l15;
    const int rcl = min(r, f - 2);
    const int shift = r - rcl;
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      const int c = r0 + 16 * (ucode[u] >> 2) + cg * 4 + l4;
      const double* colp = F + (size_t)min(c, rlim - 1) * f;      // (columns of a pivot tile: clamped inside the panel)
      d2_t v;
      __builtin_memcpy(&v, colp + rcl, 16);
      const double e0 = shift == 0 ? v[0] : v[1];
      acc[u][cg][0] = keep_f64(e0, r < rlim && c < rlim && r >= c);
      acc[u][cg][1] = keep_f64(v[1], shift == 0 && r + 1 < rlim && c < rlim && r + 1 >= c);
    }
  }
  auto await = [&](const int* state, int least, int* flag) -> bool {
    if (wave == 0) {
      int ok = 1, spins = 0;
      long long t0w = 0;
      for (;;) {
        if (__builtin_amdgcn_readfirstlane(ld_state(state)) >= least) break;
        const int stop = P.want_neg >= 0 ? __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
        const int dead = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (stop | dead) { ok = 0; break; }
        if (wait_expired(spins, t0w)) {
          if (lane == 0) atomicExch(&P.counters[5], 1ull);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *flag = ok;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*flag) != 0;
  };
  const int nrow = min(rlim - r0, 128);
  // W and L = W D^-1 of column block pb (its image and reciprocals: parity pb & 1) to memory, then the units' 8 k-steps of that block
  auto store_and_update = [&](int pb, int first_thread, int nthreads) {
    const double* Wl = Wl0 + (size_t)(pb & 1) * kWlStride;
    const double* rd2 = rdv + (pb & 1) * 32;
    if (tid >= first_thread && tid < first_thread + nthreads) {
      for (int idx = tid - first_thread; idx < 32 * 64; idx += nthreads) {
        const int p = idx >> 6, x2 = (idx & 63) * 2;
        d2_t w;
        __builtin_memcpy(&w, Wl + (size_t)p * kSyrkLd + x2, 16);
        const double rp = rd2[p];
        const d2_t l = (d2_t){w[0] * rp, w[1] * rp};
        double* wdst = Wb + (size_t)(pb * kIB + p) * f + r0 + x2;
        double* ldst = F + (size_t)(j0 + pb * kIB + p) * f + r0 + x2;
        if (x2 + 1 < nrow) { st_sc1_f64x2(wdst, w); st_sc1_f64x2(ldst, l); }
        else if (x2 < nrow) { st_agent_f64(wdst, w[0]); st_agent_f64(ldst, l[0]); }
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (!unit_on(u)) continue;
      const double* bw = Wl + 32 * (ucode[u] & 3) + 2 * l15;
      const double* bl = Wl + 16 * (ucode[u] >> 2) + (lane & 3);
      double nrd, nav[4];
      d2_t nbv;
      auto fetch = [&](int kk) {
        nrd = rd2[kk * 4 + l4];
        __builtin_memcpy(&nbv, bw + (kk * 4 + l4) * kSyrkLd, 16);
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) nav[cg] = bl[(kk * 4 + l4) * kSyrkLd + cg * 4];
      };
      fetch(0);
#pragma unroll
      for (int kk = 0; kk < kIB / 4; ++kk) {
        double av[4];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const d2_t bv = nbv;
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) av[cg] = nav[cg] * nrd;
        fetch(min(kk + 1, kIB / 4 - 1));
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
          acc[u][cg][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[cg], bv[0], acc[u][cg][0], 0, 0, 1 /* neg A */);
          acc[u][cg][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[cg], bv[1], acc[u][cg][1], 0, 0, 1 /* neg A */);
        }
      }
    }
  };
  // X = inv(L(b, b)) by one wave: diag2_body's tail, operation for operation (lane c keeps the even rows of column c, lane c + 32 the odd
  // ones; x[p] crosses with v_permlane32_swap; column p + 1 of L requested before the FMAs of column p)
  auto invert_block = [&]() {
    const int c = lane & 31, hf = lane >> 5;
    const double* Lh = Li + hf;
    double v[16], la[16], lb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (2 * j + hf == c) ? 1.0 : 0.0;
    auto both = [&](double x, int owner) {
      const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
      const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto bq = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      return owner == 0 ? __hiloint2double((int)bq[0], (int)a[0]) : __hiloint2double((int)bq[1], (int)a[1]);
    };
#pragma unroll
    for (int j = 0; j < 16; ++j) la[j] = Lh[2 * j];
#pragma unroll
    for (int p = 0; p < 32; p += 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = (p + 2) / 2; j < 16; ++j) lb[j] = Lh[2 * j + (p + 1) * kXld];
      {
        const double mine = (p >= c) ? v[p / 2] : 0.0;
        const double xp = both(mine, 0);
        v[p / 2] = hf == 0 ? xp : __builtin_fma(-la[p / 2], xp, v[p / 2]);
#pragma unroll
        for (int j = p / 2 + 1; j < 16; ++j) v[j] = __builtin_fma(-la[j], xp, v[j]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (p + 2 < 32) {
#pragma unroll
        for (int j = (p + 2) / 2; j < 16; ++j) la[j] = Lh[2 * j + (p + 2) * kXld];
      }
      {
        const double mine = (p + 1 >= c) ? v[p / 2] : 0.0;
        const double xp = both(mine, 1);
        v[p / 2] = hf == 1 ? xp : v[p / 2];
#pragma unroll
        for (int j = p / 2 + 1; j < 16; ++j) v[j] = __builtin_fma(-lb[j], xp, v[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) Xi[(2 * j + hf) + c * kXld] = v[j];
  };
  auto block_step = [&](auto bc) -> bool {
    constexpr int b = decltype(bc)::value;
    // (the barrier inside the wait also separates this block's staging from the previous block's readers of the same LDS)
    if (!await(sub_state, b + 1, s_flag + (b & 1))) return false;
    if (marks && tid == 0 && b == 0) marks[0] = wall_clock64();             // the first column block of D(q) has arrived
    if (marks && tid == 0 && b == NBLK - 1) marks[1] = wall_clock64();      // the last one: L(q, q) and D are complete
    {
      const int e = tid * 2;                 // 2 consecutive rows of one column per thread and block
      const int cc = e / kIB, rr = e - cc * kIB;
      const int gc = b * kIB + cc;
#pragma unroll
      for (int bi = b; bi < NBLK; ++bi) {
        const int gr = bi * kIB + rr;
        const double* src = F + (size_t)(j0 + gc) * f + j0 + gr;
        const double v0 = src[0], v1 = src[1];
        if (bi == b) { Li[rr + cc * kXld] = v0; Li[rr + 1 + cc * kXld] = v1; }      // (entries on and above the diagonal are never read)
        else { const d2_t v = (d2_t){v0, v1}; __builtin_memcpy(Sb + (size_t)(bi - b - 1) * kIB * kIB + e, &v, 16); }
      }
      if (tid < kIB) rdv[(b & 1) * 32 + tid] = 1.0 / P.dvals[col0 + j0 + b * kIB + tid];
    }
    __syncthreads();
    // wave 7 inverts; the others send block b - 1 on its way and apply it to the diagonal tile
    if (wave == 7) invert_block();
    else if (b > 0) store_and_update(b - 1, 0, 7 * 64);
    __syncthreads();
    {
      // W_b = t_b X_bb^T (X_bb lower triangular), then the later blocks lose W_b L_b'b^T
      double wt[8];
#pragma unroll
      for (int gp = 0; gp < 8; ++gp) {
        wt[gp] = 0.0;
#pragma unroll
        for (int g = 0; g <= gp; ++g)
          wt[gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Xi[(gp * 4 + li) + (g * 4 + lk) * kXld], t[b * 8 + g], wt[gp], 0, 0, 0);
      }
#pragma unroll
      for (int gp = 0; gp < 8; ++gp) t[b * 8 + gp] = wt[gp];
      double* Wl = Wl0 + (size_t)(b & 1) * kWlStride;
#pragma unroll
      for (int gp = 0; gp < 8; ++gp) Wl[(size_t)(4 * gp + lk) * kSyrkLd + trow] = t[b * 8 + gp];      // rows past the block row are zero
#pragma unroll
      for (int bi = b + 1; bi < NBLK; ++bi) {
        const double* Lb = Sb + (size_t)(bi - b - 1) * kIB * kIB;
#pragma unroll
        for (int gp = 0; gp < 8; ++gp)
#pragma unroll
          for (int g = 0; g < 8; ++g)
            t[bi * 8 + gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Lb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[b * 8 + g], t[bi * 8 + gp], 0, 0, 1 /* neg A */);
      }
    }
    return true;
  };
  if (!block_step(std::integral_constant<int, 0>())) return false;
  if (!block_step(std::integral_constant<int, 1>())) return false;
  if (!block_step(std::integral_constant<int, 2>())) return false;
  if (!block_step(std::integral_constant<int, 3>())) return false;
  __syncthreads();                                         // the last W block is in its image
  if (marks && tid == 0) marks[2] = wall_clock64();        // rows solved
  store_and_update(NBLK - 1, 0, kDfThreadsC);
  if (marks && tid == 0) marks[3] = wall_clock64();        // diagonal tile updated
  if (!with_d) {
    // the diagonal tile to memory: it is what D(q + 1) waits for.  Every storing wave drains, the workgroup meets, one wave publishes
    // (the block row's W and L left block by block above: the worker's publication of tile (q + 1, q) drains them)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (!unit_on(u)) continue;
      const int r = r0 + 32 * (ucode[u] & 3) + 2 * l15;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        const int c = r0 + 16 * (ucode[u] >> 2) + cg * 4 + l4;
        if (c >= rlim) continue;
        double* colp = F + (size_t)c * f;
        if (r + 1 < rlim && r >= c) {
          st_sc1_f64x2(colp + r, (d2_t){acc[u][cg][0], acc[u][cg][1]});
        } else {
          if (r < rlim && r >= c) st_agent_f64(colp + r, acc[u][cg][0]);
          if (r + 1 < rlim && r + 1 >= c) st_agent_f64(colp + r + 1, acc[u][cg][1]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) __hip_atomic_store(diag_state, dval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // tile (q + 1, q + 1) has received panel q
  } else {
    // TU(q) + D(q + 1) in one task: the updated tile is handed to the diagonal-block factorisation through LDS (column-major, leading
    // dimension kDfTileLd, over everything this function kept there)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                     // every wave has read its part of the W image
    double* Tl = sm;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (!unit_on(u)) continue;
      const int r = 32 * (ucode[u] & 3) + 2 * l15;
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        const int c = 16 * (ucode[u] >> 2) + cg * 4 + l4;
        const d2_t v = (d2_t){acc[u][cg][0], acc[u][cg][1]};
        __builtin_memcpy(Tl + (size_t)c * kDfTileLd + r, &v, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) __hip_atomic_store(row_state, dval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // tile (q + 1, q): W and L of the block row are in memory
  }
  return true;
}

// a double rotated by CTRL - 0x120 lanes inside every row of 16 lanes (DPP row_ror on the two halves: VALU, no LDS)
template <int CTRL>
__device__ __forceinline__ double df_row_ror(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// ---- U: tiles (i .. i + R - 1, j) -= W[rows, j0 .. j0 + nb) * L[columns of block j, j0 .. j0 + nb)^T ------------------------------
// k_big_syrk's tile (numeric.hip: 128 x 128 per workgroup of 2 x 4 waves, v_mfma_f64_4x4x4 with neg-A, both operand panels through
// an LDS-DMA ring of two 16-column chunks) with a row limit (the last pivot block of a front may be shorter than 128 rows) and sc1
// stores of the C tile.  One workgroup per CU has nobody to hide a tile's prologue (C tile: 6 us until it has landed, 4 us for the
// first operand chunk behind it) and epilogue, so a bulk task carries R row tiles of one tile column as ONE stream of operand
// chunks: the C tile of the next row block is requested while the current one is in its main loop (raw pairs in registers, masked
// when the accumulators switch), its first operand chunk follows the last chunk of the current tile through the ring, and the
// stores of a finished tile drain behind the next tile's first chunk.
// kDfRot (round 6): the loop is bound by the CU's LDS pipe (12 x ds_read_b64 per k-step and wave, eight waves, one pipe).  A column fragment
// holds 16 distinct doubles (4 columns x 4 k) replicated over the four blocks of v_mfma_f64_4x4x4; instead ONE read fetches 16 columns x 4 k
// (one double per lane) and the other three operands are that register rotated by 4 / 8 / 12 lanes inside every row of 16 lanes (DPP, VALU):
// block t of MFMA m multiplies with column group (t -+ m) & 3 -- the same 16 x 16 products in the same k order, dealt to the accumulators
// in a skewed order that col_of() below (C tile loads, masks, stores) follows.  6 reads per k-step; scripts/lds_dpp_probe.hip.
// Measured (MI355X): bitwise the same factor (tests/test_gpu_dataflow.py with -DOKKT_DF_ROT=1) and SLOWER -- the probe's loop 21.8 instead of
// 19.0 cycles per MFMA and SIMD, S-metric 19.2 instead of 17.1 ms: the 12 v_mov_b32_dpp of a k-step do not hide behind the MFMAs, each costs
// the wave ~ 7.5 issue cycles, more than the LDS read it replaces.  The loop is bound by what a pair of waves can ISSUE per MFMA slot (LDS
// reads and VALU alike), which is why the wider register tile (df_syrk_macro: fewer operand instructions per MFMA) helps and this does not.  Off.
// KC / STAGES: panel columns per ring slot and slots (32 x 2 for the one-workgroup-per-CU worker; 16 x 2 = 72 KB was the 128-VGPR
// bulk kernel's of the two-kernel form, scripts/experiments/r05_two_kernel_form.patch); MULTI: a task may carry several row tiles
// (the next C tile in a second register set); STAGGER: see below.
template <int KC, int STAGES, bool MULTI, bool STAGGER>
__device__ __forceinline__ void df_syrk_tiles(const DevPlan& P, int s, int j0, int nb, int i, int R_, int j, int KB, int k, double* sm, long long* marks, int* pub0 = nullptr, int pub_stride = 0, int pub_val = 0) {
  constexpr int NW = kSyrkNW;
  constexpr int kDfKC = KC;
  const int R = MULTI ? R_ : 1;
  constexpr int DMA = 2 * (kDfKC / NW);   // LDS-DMA instructions per wave and chunk
  constexpr int WCW = 128 / (NW / 2);   // columns per wave
  constexpr int NCG = WCW / 4;          // 4-column groups per wave
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // opaque: nothing derived from the lane id is hoisted out of the worker's loop and kept live across the other roles
  const int tid = tid_, lane = tid & 63, wv = tid >> 6;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int ct0 = df_block_lo(j, KB, k, f), clim = df_block_lo(j + 1, KB, k, f);
  const int cbase = ct0 + (wv >> 1) * WCW;
  double* F = P.arena + P.front_pos[s];
  double* Fc = F + (j >= KB ? P.cb_shift[s] : 0);      // the tile column's base: a block of the contribution block lives in the shared region
  const double* Wcol = P.wbuf + P.wbuf_pos[s] + (size_t)j0 * f + lane * 2;
  const double* Lg = F + (size_t)j0 * f + ct0 + lane * 2;
  const int l15 = lane & 15, l4 = lane >> 4;
  // kDfRot: the column group (of four) that MFMA m of a half multiplies with in this lane's block -- whatever the rotation brought there
  const int quad = (lane >> 2) & 3;
  int grp1 = quad, grp2 = quad, grp3 = quad;
  if constexpr (kDfRot) {
    grp1 = __builtin_amdgcn_update_dpp(0, quad, 0x124, 0xf, 0xf, false);
    grp2 = __builtin_amdgcn_update_dpp(0, quad, 0x128, 0xf, 0xf, false);
    grp3 = __builtin_amdgcn_update_dpp(0, quad, 0x12C, 0xf, 0xf, false);
  }
  auto col_of = [&](int cg) {      // the column of accumulator group cg in this lane
    if constexpr (!kDfRot) return cbase + cg * 4 + l4;
    const int m = cg & 3;
    return cbase + (cg >> 2) * 16 + (m == 0 ? quad : m == 1 ? grp1 : m == 2 ? grp2 : grp3) * 4 + l4;
  };
  const int nchunk = (nb + kDfKC - 1) / kDfKC;
  const int total = R * nchunk;
  // operand chunk g of the stream: chunk g % nchunk of row tile g / nchunk
  auto issue = [&](int g) {
    const int r = g / nchunk, ch = g - r * nchunk;
    const double* Wg = Wcol + df_block_lo(i + r, KB, k, f);
    double* slot = sm + (size_t)(g % STAGES) * 2 * kDfKC * kSyrkLd;
#pragma unroll
    for (int qq = 0; qq < kDfKC / NW; ++qq) {
      const int prow = qq * NW + wv;
      const int p = ch * kDfKC + prow;
      const double* wsrc = p < nb ? Wg + (size_t)p * f : P.zero_page + lane * 2;
      const double* lsrc = p < nb ? Lg + (size_t)p * f : P.zero_page + lane * 2;
      __builtin_amdgcn_global_load_lds(wsrc, (lds_void_t*)(slot + prow * kSyrkLd), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(lsrc, (lds_void_t*)(slot + (kDfKC + prow) * kSyrkLd), 16, 0, 0);
    }
  };
  // raw C pairs of a row tile (clamped addresses, no branches) and the masks that turn them into accumulators
  d2_t raw[NCG][2];
  auto load_c = [&](int r) {
    const int rbase = df_block_lo(i + r, KB, k, f) + (wv & 1) * 64;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      const int c = col_of(cg);
      const double* colp = Fc + (size_t)min(c, clim - 1) * f;      // (clamped inside the tile column: a pivot block never reads past its panel)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = rbase + 2 * l15 + 32 * h;
        __builtin_memcpy(&raw[cg][h], colp + min(rr, f - 2), 16);
      }
    }
  };
  double acc[NCG][4];
  auto mask_c = [&](int r) {
    const int rbase = df_block_lo(i + r, KB, k, f) + (wv & 1) * 64;
    const int rlim = df_block_lo(i + r + 1, KB, k, f);
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      const int c = col_of(cg);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = rbase + 2 * l15 + 32 * h;
        const int shift = rr - min(rr, f - 2);      // 0 in the interior, 1 when rr is the last row, >= 2 outside
        const double e0 = shift == 0 ? raw[cg][h][0] : raw[cg][h][1];
        acc[cg][2 * h] = keep_f64(e0, rr < rlim && c < clim && rr >= c);
        acc[cg][2 * h + 1] = keep_f64(raw[cg][h][1], shift == 0 && rr + 1 < rlim && c < clim && rr + 1 >= c);
      }
    }
  };
  load_c(0);
  asm volatile("" ::: "memory");
  mask_c(0);
#pragma unroll
  for (int g = 0; g < STAGES - 1; ++g) if (g < total) issue(g);
  // pub0 (round 6): a row tile of a multi-tile task is published as soon as its stores have drained -- behind the wait of the first chunk
  // iteration of the NEXT row tile -- instead of with the whole task (the caller then publishes the last row tile only): the tiles of a pair
  // used to become visible one tile's time late
  int pend_r = -1;
  int plain_until = 0;                     // chunks up to this one are waited for with vmcnt(0): ordinary loads / stores sit between the operand requests
  const int pre = min(2, nchunk - 1);      // chunk of a tile behind whose operand request the next tile's C is requested
  int r = 0, ch = 0;
  for (int g = 0; g < total; ++g) {
    {
      // chunk g has landed once only the chunks requested after it (at most STAGES - 2) are still outstanding
      const int later = g < plain_until ? 0 : min(STAGES - 2, total - 1 - g);
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (MULTI) {
      if (pend_r >= 0) {      // every wave's stores of that row tile have drained behind this iteration's vmcnt(0)
        if (tid == 0) __hip_atomic_store(pub0 + (size_t)pend_r * pub_stride, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_r = -1;
      }
    }
    if (marks && g == 0 && tid == 0) { marks[0] = wall_clock64(); marks[3] = -(long long)clock64(); }       // C tile and the first operand chunk have landed; shader clock at the start of the main loop
    // the slot written next was last read one iteration ago; everyone is past that barrier.  Waves k and k + 4 share a SIMD and run
    // the same program: waves 0 - 3 request the next chunk right here, waves 4 - 7 in the middle of their MFMAs, so that one wave's
    // LDS-DMA issue (8 requests with their address arithmetic: ~ 1 200 cycles of a 10 500-cycle chunk) falls into its partner's
    // matrix work instead of both stalling the SIMD's matrix pipe at the same time (MI355X guide, two waves per SIMD, item 9)
    const bool late = STAGGER && wv >= 4;
    const bool more = g + STAGES - 1 < total;
    if (more && !late) issue(g + STAGES - 1);
    if constexpr (MULTI) { if (ch == pre && r + 1 < R) { load_c(r + 1); plain_until = g + STAGES; } }
    const int rt0 = df_block_lo(i + r, KB, k, f), rlim = df_block_lo(i + r + 1, KB, k, f);
    const int rbase = rt0 + (wv & 1) * 64;
    // (timing experiment, OKKT_DEBUG_DF_HALF: a bulk task skips the products of every other chunk -- WRONG numbers, the operand stream and everything
    // around the loop unchanged: does the launch get faster when the bulk work gets cheaper?)
    const bool active = !(rbase + 63 < cbase) && rbase < rlim && cbase < clim && !(kDfChain && P.df_dbg_half && nb >= 256 && (g & 1));
    if (active) {
      const double* slot = sm + (size_t)(g % STAGES) * 2 * kDfKC * kSyrkLd;
      const double* bw = slot + (wv & 1) * 64 + 2 * l15;
      const double* bl = slot + kDfKC * kSyrkLd + (wv >> 1) * WCW + (kDfRot ? l15 : (lane & 3));
      if constexpr (kDfRot || !kDfPipe) {
  #pragma unroll
        for (int kk = 0; kk < kDfKC / 4; ++kk) {
          if (kk == (KC / 8) && more && late) issue(g + STAGES - 1);
          double bv[4];
  #pragma unroll
          for (int rb = 0; rb < 4; ++rb) bv[rb] = bw[(kk * 4 + l4) * kSyrkLd + (rb & 1) + 32 * (rb >> 1)];
  #pragma unroll
          for (int half = 0; half < NCG / 4; ++half) {
            double av[4];
  #pragma unroll
            for (int qq = 0; qq < 4; ++qq) if (!kDfRot || qq == 0) av[qq] = bl[(kk * 4 + l4) * kSyrkLd + (half * 4 + qq) * 4];
            if constexpr (kDfRot) {      // 16 columns x 4 k in one read; the other three fragments are that register rotated inside the rows of 16 lanes
              av[1] = df_row_ror<0x124>(av[0]);
              av[2] = df_row_ror<0x128>(av[0]);
              av[3] = df_row_ror<0x12C>(av[0]);
            }
  #pragma unroll
            for (int qq = 0; qq < 4; ++qq)
  #pragma unroll
              for (int rb = 0; rb < 4; ++rb)
                acc[half * 4 + qq][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[qq], bv[rb], acc[half * 4 + qq][rb], 0, 0, 1 /* neg A */);
          }
        }
      } else {
        // kDfPipe (round 6): the operand fragments of step st + 1 (a k-step half: four column fragments, every other step the four row fragments of
        // the next k-step too) are requested BEFORE the sixteen MFMAs of step st, in a second register set: hipcc's own schedule requests a step's
        // fragments right in front of its MFMAs and waits lgkmcnt(0) -- fine while the SIMD's other wave has MFMAs to issue, but behind every chunk
        // barrier both waves of a SIMD stand at the same reads.  scripts/lds_dpp_probe.hip (MODE 3): 19.3 instead of 21.7 cycles per MFMA and SIMD
        // with the operand stream running, 17.8 / 19.0 without.  IN THE KERNEL (-DOKKT_DF_PIPE=1: bitwise the same factor) the chunk takes 10 547 cycles
        // against 10 355 with hipcc's own order, S-metric / S-C3 / S-C5 unchanged: under the real operand stream and at the 2.0 GHz the chip runs at
        // under this load, the loop is not waiting for its LDS reads.  Off.
        static_assert(NCG == 8, "two halves of four column groups per k-step");
        // The reads and their waits are written by hand: with LDS-DMA in the loop hipcc waits lgkmcnt(0) for reads it has just issued.  LDS returns
        // in order, so lgkmcnt(2) / lgkmcnt(4) in front of a step's MFMAs means "everything but the reads just requested" (two read2_b64 for the
        // next step's column fragments, two more b128 when it starts a new k-step); the waits carry the step's operands as in / out registers so that
        // its MFMAs stay behind them, and a sched_barrier closes every step.
        const unsigned ba = (unsigned)(size_t)(bl + l4 * kSyrkLd);      // LDS byte addresses of the column / row fragments of k-step 0
        const unsigned bb = (unsigned)(size_t)(bw + l4 * kSyrkLd);
        d2_t B0[2], B1[2], A0[2], A1[2];      // two buffers each: [0] = values (0, 1), [1] = values (2, 3)
        auto ldb = [&](int kk, d2_t (&bq)[2]) {
          const unsigned ad = bb + kk * 4 * kSyrkLd * 8;
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:256" : "=&v"(bq[0]), "=&v"(bq[1]) : "v"(ad));
        };
        auto lda = [&](int kk, int half, d2_t (&aq)[2]) {
          const unsigned ad = ba + kk * 4 * kSyrkLd * 8 + half * 128;
          asm volatile("ds_read2_b64 %0, %2 offset1:4\n\tds_read2_b64 %1, %2 offset0:8 offset1:12" : "=&v"(aq[0]), "=&v"(aq[1]) : "v"(ad));
        };
        ldb(0, B0);
        lda(0, 0, A0);
#pragma unroll
        for (int st = 0; st < 2 * (kDfKC / 4); ++st) {
          const int kk = st >> 1, half = st & 1;
          d2_t (&Ac)[2] = (st & 1) ? A1 : A0;
          d2_t (&Bc)[2] = (kk & 1) ? B1 : B0;
          if (st == 2 * (KC / 8) && more && late) issue(g + STAGES - 1);
          if (st + 1 < 2 * (kDfKC / 4)) {
            const int k2 = (st + 1) >> 1, h2 = (st + 1) & 1;
            if (h2 == 0) ldb(k2, (k2 & 1) ? B1 : B0);
            lda(k2, h2, ((st + 1) & 1) ? A1 : A0);
          }
          if (st + 1 >= 2 * (kDfKC / 4)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Bc[0]), "+v"(Bc[1]));
          else if (((st + 1) & 1) == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Bc[0]), "+v"(Bc[1]));
          else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Bc[0]), "+v"(Bc[1]));
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
              acc[half * 4 + qq][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ac[qq >> 1][qq & 1], Bc[rb >> 1][rb & 1], acc[half * 4 + qq][rb], 0, 0, 1 /* neg A */);
          __builtin_amdgcn_sched_barrier(0);      // nothing moves across the step boundary
        }
      }
    } else if (more && late) issue(g + STAGES - 1);      // a wave without a piece of this tile still stages its rows
    if (++ch == nchunk) {
      if (marks && g == total - 1 && tid == 0) { marks[1] = wall_clock64(); marks[3] += (long long)clock64(); }      // main loop done: marks[3] = shader cycles of the main loop
      // the row tile is done: store it (write-through, not waited for here) and switch to the next one's accumulators
      if (active) {
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          const int c = col_of(cg);
          if (c >= clim) continue;
          double* colp = Fc + (size_t)c * f;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int rr = rbase + 2 * l15 + 32 * h;
            if (rr + 1 < rlim && rr >= c) {
              st_sc1_f64x2(colp + rr, (d2_t){acc[cg][2 * h], acc[cg][2 * h + 1]});
            } else {
              if (rr < rlim && rr >= c) st_agent_f64(colp + rr, acc[cg][2 * h]);
              if (rr + 1 < rlim && rr + 1 >= c) st_agent_f64(colp + rr + 1, acc[cg][2 * h + 1]);
            }
          }
        }
      }
      ch = 0;
      plain_until = g + STAGES;
      if constexpr (MULTI) { if (pub0 && r + 1 < R) pend_r = r; }
      ++r;
      if constexpr (MULTI) { if (r < R) mask_c(r); }
    }
  }
  if (marks && tid == 0) marks[2] = wall_clock64();        // stores issued
  // the ring's slots are reused by the next task of this workgroup: every wave is done reading them behind the caller's barrier
}

// ---- U on a MACRO tile (round 6): the tiles (i, j) and (i + 1, j), i > j, of one tile column in ONE pass --------------------------------
// What bounds the single-tile loop above: not the matrix pipe (80 % of its issue rate) and not the operand stream (it arrives a chunk
// ahead) but the LDS PIPE -- a wave reads 12 doubles (4 row fragments + 8 column fragments) per k-step of 32 MFMAs, eight waves, 4 cycles
// per 64-lane b64 read: 384 of the 512 cycles the matrix pipe needs for the same k-step, bank conflicts (0.14) on top.  The register tile
// per wave is what sets that ratio.  Here a wave owns 128 rows x 32 columns -- one whole row tile of the pair and a quarter of the tile
// column: 8 + 8 fragments per 64 MFMAs, 256 LDS cycles per 512 -- with 64 accumulators (128 VGPRs; the worker has 256).  The operand ring
// holds both W tiles and the L tile, 16 panel columns per slot (3 x 16 x 144 doubles, two slots: 108 KB).  Every entry receives its
// products in ascending k as in the single-tile task: bitwise the same tiles.  The queue builder hands out pairs only while a front is
// in its update-bound phase (dataflow_sched.cpp, OKKT_DF_ROWS_BIG / _MINKB / _AHEAD).
constexpr int kMacroKC = 16;
constexpr size_t kDfMacroLds = (size_t)2 * 3 * kMacroKC * kSyrkLd * sizeof(double);
__device__ __forceinline__ void df_syrk_macro(const DevPlan& P, int s, int j0, int nb, int i, int j, int KB, int k, double* sm, long long* marks) {
  constexpr int KC = kMacroKC, STAGES = 2;
  constexpr int DMA = 3 * KC / 8;        // LDS-DMA instructions per wave and chunk (48 one-KiB rows over eight waves)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wv = tid >> 6;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int ct0 = df_block_lo(j, KB, k, f), clim = df_block_lo(j + 1, KB, k, f);
  const int rt = wv & 1;                                   // the row tile of the pair this wave owns
  const int cbase = ct0 + (wv >> 1) * 32;
  const int rt0 = df_block_lo(i + rt, KB, k, f), rlim = df_block_lo(i + rt + 1, KB, k, f);
  double* F = P.arena + P.front_pos[s];
  double* Fc = F + (j >= KB ? P.cb_shift[s] : 0);
  const double* Wcol = P.wbuf + P.wbuf_pos[s] + (size_t)j0 * f + lane * 2;
  const double* Lg = F + (size_t)j0 * f + ct0 + lane * 2;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nchunk = (nb + KC - 1) / KC;
  const int r0a = df_block_lo(i, KB, k, f), r0b = df_block_lo(i + 1, KB, k, f);
  // slot layout: [3 arrays: W of tile i, W of tile i + 1, L][KC panel columns][kSyrkLd]
  static_assert(KC == 16, "the request pattern below deals 48 rows to eight waves: rows wv and 8 + wv of each of the three arrays");
  const double* wa = Wcol + r0a;
  const double* wb = Wcol + r0b;
  const double* zp = P.zero_page + lane * 2;
  auto issue = [&](int g) {
    double* slot = sm + (size_t)(g % STAGES) * 3 * KC * kSyrkLd;
#pragma unroll
    for (int qq = 0; qq < DMA; ++qq) {
      const int prow = (qq & 1) * 8 + wv;                  // panel column inside the chunk; array qq / 2
      const int p = g * KC + prow;
      const double* base = qq < 2 ? wa : (qq < 4 ? wb : Lg);
      const double* in = base + (size_t)min(p, nb - 1) * f;
      const double* src = p < nb ? in : zp;
      __builtin_amdgcn_global_load_lds(src, (lds_void_t*)(slot + (size_t)((qq >> 1) * KC + prow) * kSyrkLd), 16, 0, 0);
    }
  };
  // the wave's C block: 128 rows x 32 columns, pairs of rows (2 l15 + 32 h, + 1), column cbase + 4 cg + l4 -- loaded straight into the
  // accumulators (clamped addresses, no branches) and masked in place
  d2_t acc[8][4];
#pragma unroll
  for (int cg = 0; cg < 8; ++cg) {
    const int c = cbase + cg * 4 + l4;
    const double* colp = Fc + (size_t)min(c, clim - 1) * f;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int rr = rt0 + 2 * l15 + 32 * h;
      __builtin_memcpy(&acc[cg][h], colp + min(rr, f - 2), 16);
    }
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int cg = 0; cg < 8; ++cg) {
    const int c = cbase + cg * 4 + l4;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int rr = rt0 + 2 * l15 + 32 * h;
      const int shift = rr - min(rr, f - 2);      // 0 in the interior, 1 when rr is the last row, >= 2 outside
      const double e0 = shift == 0 ? acc[cg][h][0] : acc[cg][h][1];
      const double e1 = acc[cg][h][1];
      acc[cg][h][0] = keep_f64(e0, rr < rlim && c < clim);
      acc[cg][h][1] = keep_f64(e1, shift == 0 && rr + 1 < rlim && c < clim);
    }
  }
  issue(0);
  const bool active = rt0 < rlim && cbase < clim;
  for (int g = 0; g < nchunk; ++g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk g has landed (the next one is requested behind the barrier)
    __builtin_amdgcn_s_barrier();
    if (marks && g == 0 && tid == 0) { marks[0] = wall_clock64(); marks[3] = -(long long)clock64(); }
    const bool late = kDfStagger && wv >= 4;
    const bool more = g + 1 < nchunk;
    if (more && !late) issue(g + 1);
    if (active) {
      const double* slot = sm + (size_t)(g % STAGES) * 3 * KC * kSyrkLd;
      const double* bw = slot + (size_t)rt * KC * kSyrkLd + 2 * l15;
      const double* bl = slot + (size_t)2 * KC * kSyrkLd + (wv >> 1) * 32 + (lane & 3);
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        if (kk == KC / 8 && more && late) issue(g + 1);
        double bv[8];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) bv[rb] = bw[(kk * 4 + l4) * kSyrkLd + (rb & 1) + 32 * (rb >> 1)];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          double av[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) av[qq] = bl[(kk * 4 + l4) * kSyrkLd + (half * 4 + qq) * 4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int rb = 0; rb < 8; ++rb)
              acc[half * 4 + qq][rb >> 1][rb & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[qq], bv[rb], acc[half * 4 + qq][rb >> 1][rb & 1], 0, 0, 1 /* neg A */);
        }
      }
    } else if (more && late) issue(g + 1);
  }
  if (marks && tid == 0) { marks[1] = wall_clock64(); marks[3] += (long long)clock64(); }
  if (active) {
#pragma unroll
    for (int cg = 0; cg < 8; ++cg) {
      const int c = cbase + cg * 4 + l4;
      if (c >= clim) continue;
      double* colp = Fc + (size_t)c * f;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int rr = rt0 + 2 * l15 + 32 * h;
        if (rr + 1 < rlim) st_sc1_f64x2(colp + rr, acc[cg][h]);
        else if (rr < rlim) st_agent_f64(colp + rr, acc[cg][h][0]);
      }
    }
  }
  if (marks && tid == 0) marks[2] = wall_clock64();
}

// ---- U tasks CHAINED (round 6): a worker streams bulk update tasks without leaving the operand ring --------------------------------------
// The factorisation's time at S-metric is 6.6 ms + 2 740 worker-ms / workers (measured with 96 ... 256 workers: scripts/r06_workers.sh): 62 % of
// it is the workers' THROUGHPUT, and an update task spends 11 of its 91 us outside the MFMA loop -- pop and readiness 2.1, C tile and first
// operand chunk 5.2, stores 2.6, drain and publication 1.3.  Two or four row tiles per task hid that (-17 % per tile, round 4) and lost more
// in the schedule: tasks twice as long, tiles published at the end of the pair.  Here the granularity stays one tile per task.  While
// a worker is in the last seven chunks of a bulk tile that lies far enough from the panel (DevPlan::df_chain block columns behind its last
// panel) it takes its next queue position early; if that task is another update of one tile with at least two panels it polls its
// tile states ONCE and, if they are there, requests the next C tile into the second register set (the
// machinery of the multi-row task) -- every step issues its loads in one chunk iteration and reads them in the next, behind the
// vmcnt(0) the ring needs anyway, so nobody waits for a round trip.  At the end of the tile the stores leave, the next tile's first
// operand chunk is already in the ring, and the finished tile is published one chunk later (its stores have drained behind that
// iteration's wait).  A task that is of another kind, or whose inputs are not there yet, is simply the worker's next task (t_claim).  Same products in the same
// order: bitwise the same tiles.
struct DfUPar {      // wave-uniform parameters of one single-tile update task
  int s, f, k, KB, TB, i, j, q0, nq, j0, nb, nchunk, ct0, clim, rt0, rlim, t;
  double* F; double* Fc; const double* Wb;
};
__device__ __forceinline__ void df_upar(const DevPlan& P, DfUPar& u, int t, int s, int i, int j, int q0, int nq) {
  u.t = t; u.s = s; u.i = i; u.j = j; u.q0 = q0; u.nq = nq;
  u.k = P.sn_col0[s + 1] - P.sn_col0[s];
  u.f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  u.KB = (u.k + 127) >> 7;
  u.TB = u.KB + ((u.f - u.k + 127) >> 7);
  u.j0 = q0 * 128;
  u.nb = min(nq * 128, u.k - u.j0);
  u.nchunk = (u.nb + kDfKC - 1) / kDfKC;
  u.ct0 = df_block_lo(j, u.KB, u.k, u.f); u.clim = df_block_lo(j + 1, u.KB, u.k, u.f);
  u.rt0 = df_block_lo(i, u.KB, u.k, u.f); u.rlim = df_block_lo(i + 1, u.KB, u.k, u.f);
  u.F = P.arena + P.front_pos[s];
  u.Fc = u.F + (j >= u.KB ? P.cb_shift[s] : 0);
  u.Wb = P.wbuf + P.wbuf_pos[s] + (size_t)u.j0 * u.f;
}
constexpr int kChainLead = 7;      // chunk iterations between the look at the queue head and the end of the tile (five steps + the C tile + the first chunk)
// returns through t_last the queue position of the LAST tile of the chain (the caller drains and publishes it: mine / newv), through
// t_claim a claimed task that has not been started (-1: none)
__device__ __forceinline__ void df_syrk_chain(const DevPlan& P, const DfTask* __restrict__ tasks, int ntasks, int* __restrict__ head, int t0, int s0, int i0, int j0_, int q00, int nq0,
                                              int* s_nxt, double* sm, long long* tlog, int** mine, int* newv, int* t_last, int* t_claim) {
  constexpr int NW = kSyrkNW, KC = kDfKC;
  static_assert(kDfStages == 2, "the chained stream waits for every chunk with vmcnt(0): a ring of two slots");
  constexpr int WCW = 128 / (NW / 2), NCG = WCW / 4;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wv = tid >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(wv);
  const int l15 = lane & 15, l4 = lane >> 4;
  DfUPar cur, nxt;
  df_upar(P, cur, t0, s0, i0, j0_, q00, nq0);
  nxt = cur;
  auto issue = [&](const DfUPar& u, int ch, int slotid) {
    double* slot = sm + (size_t)slotid * 2 * KC * kSyrkLd;
    const double* Wg = u.Wb + u.rt0 + lane * 2;
    const double* Lg = u.F + (size_t)u.j0 * u.f + u.ct0 + lane * 2;
#pragma unroll
    for (int qq = 0; qq < KC / NW; ++qq) {
      const int prow = qq * NW + wv;
      const int p = ch * KC + prow;
      const double* wsrc = p < u.nb ? Wg + (size_t)p * u.f : P.zero_page + lane * 2;
      const double* lsrc = p < u.nb ? Lg + (size_t)p * u.f : P.zero_page + lane * 2;
      __builtin_amdgcn_global_load_lds(wsrc, (lds_void_t*)(slot + prow * kSyrkLd), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(lsrc, (lds_void_t*)(slot + (KC + prow) * kSyrkLd), 16, 0, 0);
    }
  };
  d2_t raw[NCG][2];
  auto load_c = [&](const DfUPar& u) {
    const int rbase = u.rt0 + (wv & 1) * 64, cbase = u.ct0 + (wv >> 1) * WCW;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      const int c = cbase + cg * 4 + l4;
      const double* colp = u.Fc + (size_t)min(c, u.clim - 1) * u.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = rbase + 2 * l15 + 32 * h;
        __builtin_memcpy(&raw[cg][h], colp + min(rr, u.f - 2), 16);
      }
    }
  };
  double acc[NCG][4];
  auto mask_c = [&](const DfUPar& u) {
    const int rbase = u.rt0 + (wv & 1) * 64, cbase = u.ct0 + (wv >> 1) * WCW;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      const int c = cbase + cg * 4 + l4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = rbase + 2 * l15 + 32 * h;
        const int shift = rr - min(rr, u.f - 2);
        const double e0 = shift == 0 ? raw[cg][h][0] : raw[cg][h][1];
        acc[cg][2 * h] = keep_f64(e0, rr < u.rlim && c < u.clim && rr >= c);
        acc[cg][2 * h + 1] = keep_f64(raw[cg][h][1], shift == 0 && rr + 1 < u.rlim && c < u.clim && rr + 1 >= c);
      }
    }
  };
  auto store_c = [&](const DfUPar& u) {
    const int rbase = u.rt0 + (wv & 1) * 64, cbase = u.ct0 + (wv >> 1) * WCW;
    if ((rbase + 63 < cbase) || rbase >= u.rlim || cbase >= u.clim) return;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      const int c = cbase + cg * 4 + l4;
      if (c >= u.clim) continue;
      double* colp = u.Fc + (size_t)c * u.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = rbase + 2 * l15 + 32 * h;
        if (rr + 1 < u.rlim && rr >= c) {
          st_sc1_f64x2(colp + rr, (d2_t){acc[cg][2 * h], acc[cg][2 * h + 1]});
        } else {
          if (rr < u.rlim && rr >= c) st_agent_f64(colp + rr, acc[cg][2 * h]);
          if (rr + 1 < u.rlim && rr + 1 >= c) st_agent_f64(colp + rr + 1, acc[cg][2 * h + 1]);
        }
      }
    }
  };
  if (wave == 0) s_nxt[0] = 0;
  load_c(cur);
  asm volatile("" ::: "memory");
  mask_c(cur);
  issue(cur, 0, 0);
  int G = 0, c = 0;                    // chunks since the start of the chain (ring slot = G & 1), chunk of the current tile
  int have = 0;                        // 0: no next tile; 2: its C tile is on its way, its first chunk follows the current tile's last
  int claimed = -1;                    // a claimed task whose inputs were not there at the one look
  int* pend = nullptr; int pendv = 0, pendt = -1;      // the finished tile that is published behind the next iteration's wait
  // wave 0's look at the queue, one step per chunk iteration (loads requested in one step are read in the next)
  int stage = 0, qh = 0, rs = 0, rtn = 0, rij = 0, rq0 = 0, st0 = 0, st1 = 0, st2 = 0, stopw = 0;
  int nk = 0, nf = 0; long long nsp = 0;
  for (;;) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // chunk G has landed; so has everything else this wave has asked for, and its stores have drained
    __builtin_amdgcn_s_barrier();
    // a tile looks for a successor only if it is long enough for the five steps and itself far from the panel (its own publication moves
    // one chunk back when it has one)
    const bool looks = P.df_chain > 0 && cur.nchunk >= kChainLead && c >= cur.nchunk - kChainLead && cur.j - (cur.q0 + cur.nq - 1) >= P.df_chain;
    if (pend) {      // the previous tile of the chain: every wave's stores have drained behind this iteration's wait
      if (wave == 0 && lane == 0) __hip_atomic_store(pend, pendv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tlog && tid == 0) tlog[(size_t)pendt * 8 + 2] = wall_clock64();
      pend = nullptr;
    }
    const int announced = __builtin_amdgcn_readfirstlane(s_nxt[0]);      // written by wave 0 in the previous iteration (stage 4)
    if (stage == 5) {
      stage = 6;
      if (announced == 2) {
        const int nt = __builtin_amdgcn_readfirstlane(s_nxt[1]), ns = __builtin_amdgcn_readfirstlane(s_nxt[2]), ntn = __builtin_amdgcn_readfirstlane(s_nxt[3]);
        const int nij = __builtin_amdgcn_readfirstlane(s_nxt[4]), nq0 = __builtin_amdgcn_readfirstlane(s_nxt[5]);
        df_upar(P, nxt, nt, ns, nij & 0xffff, nij >> 16, nq0, (ntn >> 8) & 255);
        have = 2;
      } else if (announced == 1) claimed = __builtin_amdgcn_readfirstlane(s_nxt[1]);
    }
    const bool late = kDfStagger && wv >= 4;
    const bool last = c + 1 == cur.nchunk;
    const bool more = !last || have == 2;
    auto issue_next = [&]() { if (!last) issue(cur, c + 1, (G + 1) & 1); else issue(nxt, 0, (G + 1) & 1); };
    if (more && !late) issue_next();
    if (have == 2 && stage == 6) { load_c(nxt); stage = 7; }      // the next C tile: plain loads, read behind a later iteration's wait
    if (wave == 0 && looks && stage < 5) {
      if (stage == 0) {
        // the worker's next queue position, taken early (what the caller does at the end of a task): from here on the task is this worker's,
        // chained or not.  (A look at the head followed by a compare-and-swap two steps later practically never succeeds: in the busy phase
        // another worker pops every 0.3 us.)
        qh = 0;
        if (lane == 0) qh = atomicAdd(head, 1);
        stage = 1;
      } else if (stage == 1) {
        qh = __builtin_amdgcn_readfirstlane(qh);
        rs = 0; rtn = 255; rij = 0; rq0 = 0;
        if (qh < ntasks) {
          const DfTask tk = tasks[qh];
          rs = tk.front; rtn = tk.type_nq; rij = tk.ij; rq0 = tk.q0;
        }
        stage = 2;
      } else if (stage == 2) {
        rs = __builtin_amdgcn_readfirstlane(rs); rtn = __builtin_amdgcn_readfirstlane(rtn); rij = __builtin_amdgcn_readfirstlane(rij); rq0 = __builtin_amdgcn_readfirstlane(rq0);
        nk = P.sn_col0[rs + 1] - P.sn_col0[rs];
        nf = (int)(P.row_ptr[rs + 1] - P.row_ptr[rs]);
        nsp = P.df_state_pos[rs];
        if (tlog && lane == 0 && qh < ntasks) { tlog[(size_t)qh * 8] = wall_clock64(); tlog[(size_t)qh * 8 + 3] = blockIdx.x; }
        stage = 3;
      } else if (stage == 3) {
        const int ty = rtn & 255, nqn = (rtn >> 8) & 255, rw = rtn >> 16;
        st0 = st1 = st2 = -1; stopw = 1;
        if (ty == kDfU && rw <= 1 && nqn >= 2) {      // another update of one tile with at least two panels: its three tile states, once
          const int in = rij & 0xffff, jn = rij >> 16, ql = rq0 + nqn - 1;
          const int KBn = (nk + 127) >> 7, TBn = KBn + ((nf - nk + 127) >> 7);
          const int* stn = P.df_state + nsp;
          st0 = ld_state(stn + (size_t)in * TBn + ql);      // row operand: block row i of the group's last panel
          st1 = ld_state(stn + (size_t)in * TBn + jn);      // the tile itself: the panels before the group applied
          st2 = ld_state(stn + (size_t)jn * TBn + ql);      // column operand
          stopw = (int)__hip_atomic_load(&P.counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | (P.want_neg >= 0 ? (int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
        }
        stage = 4;
      } else if (stage == 4) {
        const int nqn = (rtn >> 8) & 255, ql = rq0 + nqn - 1;
        const bool ready = __builtin_amdgcn_readfirstlane(stopw) == 0 && __builtin_amdgcn_readfirstlane(st0) >= ql + 1 && __builtin_amdgcn_readfirstlane(st1) >= rq0 && __builtin_amdgcn_readfirstlane(st2) >= ql + 1;
        if (ready) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          if (tlog && lane == 0) tlog[(size_t)qh * 8 + 1] = wall_clock64();
        }
        s_nxt[1] = qh; s_nxt[2] = rs; s_nxt[3] = rtn; s_nxt[4] = rij; s_nxt[5] = rq0;
        s_nxt[0] = ready ? 2 : 1;      // every lane writes the same words
        stage = 5;
      }
    }
    if (wave != 0 && looks && stage < 5) stage = stage + 1;      // the other waves count the steps along (the same iterations in every wave)
    {
      const int rbase = cur.rt0 + (wv & 1) * 64, cbase = cur.ct0 + (wv >> 1) * WCW;
      const bool active = !(rbase + 63 < cbase) && rbase < cur.rlim && cbase < cur.clim;
      if (active) {
        const double* slot = sm + (size_t)(G & 1) * 2 * KC * kSyrkLd;
        const double* bw = slot + (wv & 1) * 64 + 2 * l15;
        const double* bl = slot + KC * kSyrkLd + (wv >> 1) * WCW + (lane & 3);
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
          if (kk == (KC / 8) && more && late) issue_next();
          double bv[4];
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) bv[rb] = bw[(kk * 4 + l4) * kSyrkLd + (rb & 1) + 32 * (rb >> 1)];
#pragma unroll
          for (int half = 0; half < NCG / 4; ++half) {
            double av[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) av[qq] = bl[(kk * 4 + l4) * kSyrkLd + (half * 4 + qq) * 4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
              for (int rb = 0; rb < 4; ++rb)
                acc[half * 4 + qq][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[qq], bv[rb], acc[half * 4 + qq][rb], 0, 0, 1 /* neg A */);
          }
        }
      } else if (more && late) issue_next();
    }
    ++G;
    if (!last) { ++c; continue; }
    // the tile is done: its stores leave (write-through); either the next tile of the chain takes over the accumulators or the caller
    // drains and publishes
    store_c(cur);
    if (have != 2) break;
    pend = P.df_state + P.df_state_pos[cur.s] + (size_t)cur.i * cur.TB + cur.j; pendv = cur.q0 + cur.nq; pendt = cur.t;
    mask_c(nxt);
    cur = nxt;
    have = 0; c = 0; stage = 0;
    if (wave == 0) s_nxt[0] = 0;
  }
  *mine = P.df_state + P.df_state_pos[cur.s] + (size_t)cur.i * cur.TB + cur.j;
  *newv = cur.q0 + cur.nq;
  *t_last = cur.t;
  *t_claim = claimed;
}

constexpr int kDfThreads = kDfThreadsC;
constexpr size_t kDfLds = std::max(std::max(std::max(std::max(OKKT_DIAG2_LDS_DOUBLES(kMW) * sizeof(double), kDfTuLds), (size_t)kDfStages * 2 * kDfKC * kSyrkLd * sizeof(double)), kDfTlLds), std::max(std::max(kDfTuLockLds, kDfMacroLds), (size_t)128 * kDfTileLd * sizeof(double)));   // diag2_body's and df_tu_tile's; the other roles need less

// counters[5] = a wait ran into its bound, three seconds of wall clock (or another worker's did): every worker leaves, the factorisation
// fails with "a hand-off timed out" and the solves return NaN -- never numbers computed from tiles that had not arrived
__global__ __launch_bounds__(kDfThreads, 1) void k_front_dataflow(DevPlan P, const DfTask* __restrict__ tasks, int ntasks, int* __restrict__ head, double tol, int drop, int dbg, long long* __restrict__ tlog) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_ctl[8];
  __shared__ int s_nxt[8];      // df_syrk_chain: the next tile of a chain, announced by wave 0
  const int tid = threadIdx.x;
  if constexpr (!kDfLog) { tlog = nullptr; dbg = 0; }      // (the product library: no instrumentation in the kernel)
  // The scheduling code runs on WAVE 0 as a whole (a wave-uniform branch, every lane polls the same words): with an `if (tid == 0)`
  // around it hipcc threads the branch through the loop header, the structurizer turns the worker's loop into two nested loops and
  // lanes 1 .. 63 of wave 0 run ahead into the next s_barrier while lane 0 is masked off -- the workgroup hangs (round 4, first run)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t_ahead = -1;      // wave 0: the queue position taken while the previous task's stores were draining
  for (;;) {
    if (wave == 0) {
      int t = t_ahead;
      if (t_ahead < 0) {
        if ((tid & 63) == 0) t = atomicAdd(head, 1);
        t = __builtin_amdgcn_readfirstlane(t);
      }
      int ok = 1;
      // a retry of the delta loop whose pivot counts already decide a wrong inertia: nothing more is started (the workers leave
      // within one task's time; the launches of the levels above return at their first pop)
      if (P.want_neg >= 0 && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) ok = 0;
      if (tlog && t < ntasks && (tid & 63) == 0) { tlog[(size_t)t * 8] = wall_clock64(); tlog[(size_t)t * 8 + 3] = blockIdx.x; }
      if (t < ntasks && ok) {
        const DfTask tk = tasks[t];
        const int s = __builtin_amdgcn_readfirstlane(tk.front), type_nq = __builtin_amdgcn_readfirstlane(tk.type_nq), ij = __builtin_amdgcn_readfirstlane(tk.ij), q0 = __builtin_amdgcn_readfirstlane(tk.q0);
        const int type = type_nq & 255, nq = (type_nq >> 8) & 255, rows = max(type_nq >> 16, 1), i = ij & 0xffff, j = ij >> 16;
        const int k = P.sn_col0[s + 1] - P.sn_col0[s];
        const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
        const int KB = (k + 127) >> 7;
        const int TB = KB + ((f - k + 127) >> 7);
        const int* st = P.df_state + P.df_state_pos[s];
        // the tile states the task waits for, one per lane of the wave: (address, least value); the other lanes hold a condition
        // that is always true
        const int ln = tid & 63;
        const int* addr = st;
        int need = -(1 << 30);
        if (type == kDfD) { if (ln == 0) { addr = st + (size_t)i * TB + i; need = i; } }
        else if (type == kDfT) {
          if (ln == 0) { addr = st + (size_t)j * TB + j; need = j + 1; }
          if (ln == 1) { addr = st + (size_t)i * TB + j; need = j; }
        } else if (type == kDfTL) {      // tile (i, q) with panel q - 2 applied, block rows i and q of panel q - 1; D(q) is awaited inside the task
          if (ln == 0) { addr = st + (size_t)i * TB + j; need = j - 1; }
          if (ln == 1) { addr = st + (size_t)i * TB + (j - 1); need = j; }
          if (ln == 2) { addr = st + (size_t)j * TB + (j - 1); need = j; }
        } else if (type == kDfTU || type == kDfTA) {      // the tiles (q + 1, q) and (q + 1, q + 1), q = j; D(q) is awaited inside the task
          if (ln == 0) { addr = st + (size_t)i * TB + j; need = j; }
          if (ln == 1) { addr = st + (size_t)i * TB + i; need = j; }
        } else {
          const int ql = q0 + nq - 1;
          if (ln < rows) { addr = st + (size_t)(i + ln) * TB + ql; need = ql + 1; }                   // row operands
          else if (ln < 2 * rows) { addr = st + (size_t)(i + ln - rows) * TB + j; need = q0; }        // the tiles themselves
          else if (ln == 2 * rows) { addr = st + (size_t)j * TB + ql; need = ql + 1; }                // column operand
        }
        if (drop && ln == 0) need += 1 << 20;      // tests: a hand-off that never comes
        int spins = 0;
        long long t0w = 0;
        for (;;) {
          if (__builtin_amdgcn_ballot_w64(ld_state(addr) >= need) == ~0ull) break;
          // the stop flag of the delta loop (retries only) and the time-out word end every wait
          const int stop = P.want_neg >= 0 ? __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
          const int dead = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (stop | dead) { ok = 0; break; }
          if (wait_expired(spins, t0w)) {
            if (ln == 0) atomicExch(&P.counters[5], 1ull);
            ok = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tlog && (tid & 63) == 0) tlog[(size_t)t * 8 + 1] = wall_clock64();
      }
      s_ctl[0] = t; s_ctl[1] = ok;       // every lane of the wave writes the same two words
    }
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(s_ctl[0]);
    if (t >= ntasks || __builtin_amdgcn_readfirstlane(s_ctl[1]) == 0) return;
    const DfTask tk = tasks[t];
    const int s = __builtin_amdgcn_readfirstlane(tk.front);
    const int type_nq = __builtin_amdgcn_readfirstlane(tk.type_nq);
    const int ij = __builtin_amdgcn_readfirstlane(tk.ij);
    const int q0 = __builtin_amdgcn_readfirstlane(tk.q0);
    const int type = type_nq & 255, nq = (type_nq >> 8) & 255, rows = max(type_nq >> 16, 1), i = ij & 0xffff, j = ij >> 16;
    const int k = P.sn_col0[s + 1] - P.sn_col0[s];
    const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
    const int KB = (k + 127) >> 7;
    const int TB = KB + ((f - k + 127) >> 7);
    int* st = P.df_state + P.df_state_pos[s];
    int* mine;
    int newv, npub = 1;
    int t_done = t, t_hold = -1;      // chained update tasks: the last tile of the chain (published below), a claimed task that has not been started
    if (type == kDfD) {
      if (!(dbg & 1)) diag2_body<true, kDiag2MW, kDfDiagMfmaWaves, kDfProg>(P, s, i, 128, tol, sm, nullptr, 0, tlog ? tlog + (size_t)t * 8 + 4 : nullptr, (kDfProg && (nq & 8) && i + 1 < KB) ? st + (size_t)i * TB + i + 1 : nullptr);
      mine = st + (size_t)i * TB + i; newv = i + 1;
    } else if (type == kDfT) {
      if (!(dbg & 2)) df_trsm_tile(P, s, j, df_block_lo(i, KB, k, f), df_block_lo(i + 1, KB, k, f), sm);
      mine = st + (size_t)i * TB + j; newv = j + 1;
    } else if (type == kDfTL) {
      if (!df_tl_tile(P, s, j, df_block_lo(i, KB, k, f), df_block_lo(i + 1, KB, k, f), st + (size_t)j * TB + j, &s_ctl[2], sm, tlog ? tlog + (size_t)t * 8 + 4 : nullptr)) return;
      mine = st + (size_t)i * TB + j; newv = j + 1;
    } else if (type == kDfTU || type == kDfTA) {
      // nq & 2: D(q + 1) is part of the task (the updated diagonal tile reaches it through LDS); nq & 4: the lower half of a split block
      // row (the upper half is the TA task right before it in the queue)
      const bool with_d = type == kDfTU && (nq & 2);
      const int part = type == kDfTA ? 1 : ((nq & 4) ? 2 : 0);
      int* hs = st + (size_t)j * TB + i;          // the unused upper slot (q, q + 1): the state of the upper half
#ifndef OKKT_DF_NO_LOCK
      if (kDfProg && (nq & 8)) {      // in lockstep with D(q): the sub-states of D(q) live in the slot hs
        if (!df_tu_lock(P, s, j, df_block_lo(i, KB, k, f), df_block_lo(i + 1, KB, k, f), hs, j + 1, st + (size_t)i * TB + i, st + (size_t)i * TB + j, with_d,
                        &s_ctl[2], sm, tlog ? tlog + (size_t)t * 8 + 4 : nullptr)) return;
      } else
#endif
      if (!df_tu_tile(P, s, j, df_block_lo(i, KB, k, f), df_block_lo(i + 1, KB, k, f), st + (size_t)j * TB + j, j + 1, st + (size_t)i * TB + i, st + (size_t)i * TB + j, hs, part, with_d,
                      &s_ctl[2], sm, tlog ? tlog + (size_t)t * 8 + 4 : nullptr)) return;
      if (type == kDfTA) {
        mine = hs; newv = 2;
      } else if (with_d) {
        diag2_body<true, kDiag2MW, kDfDiagMfmaWaves, kDfProg>(P, s, i, 128, tol, sm, sm, kDfTileLd, nullptr, (kDfProg && (nq & 8) && i + 1 < KB) ? st + (size_t)i * TB + i + 1 : nullptr);
        mine = st + (size_t)i * TB + i; newv = i + 1;          // tile (q + 1, q) was published inside the task
      } else {
        mine = st + (size_t)i * TB + j; newv = j + 1;
      }
    } else {
      const int j0 = q0 * 128;
      // a pair of row tiles below the diagonal tile: one macro tile (a wave owns 128 x 32 of it); everything else tile by tile
      mine = st + (size_t)i * TB + j; newv = q0 + nq; npub = rows;
      if (!(dbg & 4)) {
        if (kDfMacro && rows == 2 && i > j && P.df_macro) df_syrk_macro(P, s, j0, min(nq * 128, k - j0), i, j, KB, k, sm, tlog ? tlog + (size_t)t * 8 + 4 : nullptr);
        else if (kDfChain && P.df_chain > 0 && rows == 1 && nq >= 2) df_syrk_chain(P, tasks, ntasks, head, t, s, i, j, q0, nq, s_nxt, sm, tlog, &mine, &newv, &t_done, &t_hold);
        else if (kDfMulti && rows > 1 && P.df_early_pub) {      // the row tiles but the last are published from inside the task, as their stores drain
          df_syrk_tiles<kDfKC, kDfStages, kDfMulti, kDfStagger>(P, s, j0, min(nq * 128, k - j0), i, rows, j, KB, k, sm, tlog ? tlog + (size_t)t * 8 + 4 : nullptr, mine, TB, newv);
          mine += (size_t)(rows - 1) * TB; npub = 1;
        } else df_syrk_tiles<kDfKC, kDfStages, kDfMulti, kDfStagger>(P, s, j0, min(nq * 128, k - j0), i, rows, j, KB, k, sm, tlog ? tlog + (size_t)t * 8 + 4 : nullptr);
      }
    }
    // the next queue position is requested now: the atomic's round trip (1 us) runs beside the drain of this task's stores
    int t_pre = 0;
    if (wave == 0 && (tid & 63) == 0 && t_hold < 0) t_pre = atomicAdd(head, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its sc1 stores have been acknowledged
    __syncthreads();                                      // ... and every wave is done with the LDS of this task and with s_ctl
    if (tlog && tid == 0) tlog[(size_t)t_done * 8 + 2] = wall_clock64();
    if (wave == 0) {      // lane r publishes row tile r of the task (one tile for every kind but the bulk updates)
      const int ln = tid & 63;
      if (ln < npub) __hip_atomic_store(mine + (size_t)ln * TB, newv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t_ahead = t_hold >= 0 ? t_hold : __builtin_amdgcn_readfirstlane(t_pre);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------

const char* df_build_flags() {
  if (kDfProg && kDfMacro && kDfChain) return kDfLog ? "lockstep=1 macro=1 chain=1 log=1" : "lockstep=1 macro=1 chain=1 log=0";
  if (!kDfProg && !kDfMacro && !kDfChain) return kDfLog ? "lockstep=0 macro=0 chain=0 log=1" : "lockstep=0 macro=0 chain=0 log=0";
  return "mixed";
}

std::string df_setup(Numeric& N) {
  DevPlan& d = N.d;
  const int ns = d.nsuper;
  N.df_tasks = nullptr; N.df_heads = nullptr; N.n_df_heads = 0; N.df_state_ints = 0;
  d.df_state = nullptr; d.df_state_pos = nullptr;
  if (!N.dataflow || N.nb != 128) return "";
  std::vector<int64_t> spos(ns, -1);
  int64_t total = 0;
  std::vector<DfTask> all;
  std::vector<DfTask> q;
  int nheads = 0;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount; }
  N.df_workers = getenv("OKKT_DF_WORKERS") ? std::max(1, atoi(getenv("OKKT_DF_WORKERS"))) : ncu;
  N.df_group = getenv("OKKT_DF_GROUP") ? std::max(1, std::min(atoi(getenv("OKKT_DF_GROUP")), 16)) : 4;
  N.df_fuse_d = getenv("OKKT_DF_FUSE_D") ? atoi(getenv("OKKT_DF_FUSE_D")) : 1;
  N.df_split_tu = getenv("OKKT_DF_SPLIT_TU") ? atoi(getenv("OKKT_DF_SPLIT_TU")) : 1;
  N.df_fuse_tl = getenv("OKKT_DF_FUSE_TL") ? atoi(getenv("OKKT_DF_FUSE_TL")) : 1;
  d.df_macro = getenv("OKKT_DF_MACRO") ? atoi(getenv("OKKT_DF_MACRO")) : 1;
  d.df_early_pub = getenv("OKKT_DF_EARLY_PUB") ? atoi(getenv("OKKT_DF_EARLY_PUB")) : 1;      // multi-tile update tasks publish every row tile as its stores drain (round 6; 0: with the task)
  // chained update tasks (df_syrk_chain): a bulk tile is followed by the next bulk tile of the queue without leaving the operand ring when that tile lies
  // at least this many block columns behind its group's last panel; 0 = off
  d.df_dbg_half = (kDfChain && getenv("OKKT_DEBUG_DF_HALF")) ? atoi(getenv("OKKT_DEBUG_DF_HALF")) : 0;
  d.df_chain = getenv("OKKT_DF_CHAIN") ? std::max(0, atoi(getenv("OKKT_DF_CHAIN"))) : 3;
  // OFF by default: built, bitwise equal, and slower (72 us between two diagonal blocks instead of 60) -- the tiles TU(q + 1) starts from arrive
  // through D(q) -> T / TL(q + 2, q) -> the lone last-panel updates of (q + 2, q + 1) and (q + 2, q + 2) about when D(q + 1) ENDS, so the
  // follower has nothing to follow and runs its four block steps (9 us each with its own inversions) behind D(q + 1); DESIGN.md section 4
  N.df_lockstep = kDfProg ? (getenv("OKKT_DF_LOCKSTEP") ? atoi(getenv("OKKT_DF_LOCKSTEP")) : 0) : 0;
  N.df_rows = (kDfMulti && getenv("OKKT_DF_ROWS")) ? std::max(1, std::min(atoi(getenv("OKKT_DF_ROWS")), 8)) : 1;
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_front_dataflow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  auto do_sched = [&](std::vector<LevelSchedule>& levels) {
    for (LevelSchedule& L : levels) {
      Segment& g = L.seg[3];
      g.df_off = -1; g.df_cnt = 0; g.df_head = -1; g.df_flops = 0;
      if (g.cnt == 0) continue;
      std::vector<DfFront> fronts;
      for (int a = 0; a < g.cnt; ++a) {
        const int s = N.sched_host[g.off + a];
        const int f = N.sn_f[s], k = N.sn_k[s];
        fronts.push_back({s, f, k});
        if (spos[s] < 0) {
          const int64_t TB = (k + 127) / 128 + (f - k + 127) / 128;
          spos[s] = total;
          total += TB * TB;
        }
        g.df_flops += (double)k * f * f - (double)k * k * f + (double)k * k * k / 3.0;
      }
      double model = 0;
      df_build_queue(fronts, N.df_workers, N.df_group, N.df_rows, N.df_fuse_d != 0, N.df_split_tu != 0 && !N.df_lockstep, q, &model, N.df_fuse_tl != 0, N.df_lockstep != 0, kDfMulti);
      g.df_off = (int64_t)all.size();
      g.df_cnt = (int)q.size();
      g.df_head = nheads++;
      all.insert(all.end(), q.begin(), q.end());
      if (getenv("OKKT_DEBUG_FRONTS")) fprintf(stderr, "okkt: dataflow level: %d fronts (largest %d x %d), %d tasks, model %.0f us\n", g.cnt, g.maxf, g.maxk, g.df_cnt, model);
    }
  };
  do_sched(N.levels);
  do_sched(N.levels_top);
  N.n_df_heads = nheads;
  N.df_state_ints = total;
  auto up = [&](const void* src, size_t bytes, void** out) -> std::string {
    void* p = nullptr;
    OKKT_HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 16)));
    N.allocations.push_back(p);
    if (bytes) OKKT_HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *out = p;
    return "";
  };
  std::string e;
  if (!(e = up(spos.data(), spos.size() * sizeof(int64_t), (void**)&d.df_state_pos)).empty()) return e;
  if (!(e = up(all.data(), all.size() * sizeof(DfTask), (void**)&N.df_tasks)).empty()) return e;
  if (total + (int64_t)nheads * kDfHeadStride >= (int64_t)1 << 31) return "dataflow: tile states beyond 2^31";
  {
    void* p = nullptr;      // tile states, then the queue heads: one fill per factorisation clears both
    const size_t bytes = ((size_t)total + (size_t)nheads * kDfHeadStride + 16) * sizeof(int);
    OKKT_HIP_TRY(hipMalloc(&p, bytes));
    N.allocations.push_back(p);
    OKKT_HIP_TRY(hipMemset(p, 0, bytes));
    // the handle's streams are non-blocking: nothing else orders this null-stream fill before the first factorisation's own clear and
    // launch (dalloc() documents the same race for its zero fills; advisor, round 5)
    OKKT_HIP_TRY(hipStreamSynchronize(nullptr));
    d.df_state = (int*)p;
    N.df_heads = d.df_state + total;
  }
  return "";
}

std::string df_launch(Numeric& N, const DevPlan& P, const Segment& g, hipStream_t st, double tol) {
  if (g.df_cnt <= 0) return "";
  static const int drop = getenv("OKKT_DEBUG_DROP_HANDOFF") ? atoi(getenv("OKKT_DEBUG_DROP_HANDOFF")) : 0;
  static const int dbg_env = getenv("OKKT_DEBUG_DATAFLOW") ? atoi(getenv("OKKT_DEBUG_DATAFLOW")) : 0;   // 1 / 2 / 4: skip the bodies of D / T / U (wrong results), 8: synchronise and report every launch, 16: per-task time stamps appended to $OKKT_DF_LOG
  static const int dbg = kDfLog ? dbg_env : (dbg_env & 8);
  static const bool warned = [] {
    if (!kDfLog && (dbg_env & ~8)) fprintf(stderr, "okkt: OKKT_DEBUG_DATAFLOW=%d: the task log and the body switches are compiled into libonephase_kkt_log.so and libonephase_kkt_exp.so only (OKKT_LIB_PATH); ignored here\n", dbg_env);
    return true;
  }();
  (void)warned;
  const int grid = std::min(g.df_cnt, N.df_workers);
  long long* tlog = nullptr;
  if (dbg & 16) { OKKT_HIP_TRY(hipMalloc((void**)&tlog, (size_t)g.df_cnt * 8 * sizeof(long long))); OKKT_HIP_TRY(hipMemsetAsync(tlog, 0, (size_t)g.df_cnt * 8 * sizeof(long long), st)); }   // debug only: per task pop / ready / end ticks (10 ns) and the worker
  hipLaunchKernelGGL(k_front_dataflow, dim3(grid), dim3(kDfThreads), kDfLds, st, P, N.df_tasks + g.df_off, g.df_cnt, N.df_heads + g.df_head * kDfHeadStride, tol, (drop & 4) ? 1 : 0, dbg, tlog);
  if (dbg & 24) {
    hipError_t e2 = hipStreamSynchronize(st);
    fprintf(stderr, "okkt: dataflow launch of %d tasks on %d workers: %s\n", g.df_cnt, grid, hipGetErrorString(e2));
  }
  if (tlog) {
    std::vector<long long> hl((size_t)g.df_cnt * 8);
    std::vector<DfTask> ht((size_t)g.df_cnt);
    OKKT_HIP_TRY(hipMemcpy(hl.data(), tlog, hl.size() * sizeof(long long), hipMemcpyDeviceToHost));
    OKKT_HIP_TRY(hipMemcpy(ht.data(), N.df_tasks + g.df_off, ht.size() * sizeof(DfTask), hipMemcpyDeviceToHost));
    (void)hipFree(tlog);
    const char* path = getenv("OKKT_DF_LOG");
    if (FILE* fp = fopen(path ? path : "/tmp/okkt_df_log.txt", "a")) {
      fprintf(fp, "# launch %d tasks %d workers (index front type i j q0 nq worker pop ready end [10 ns ticks])\n", g.df_cnt, grid);
      for (int t = 0; t < g.df_cnt; ++t)
        fprintf(fp, "%d %d %d %d %d %d %d %lld %lld %lld %lld %lld %lld %lld %lld\n", t, ht[t].front, ht[t].type_nq & 255, ht[t].ij & 0xffff, ht[t].ij >> 16, ht[t].q0, ht[t].type_nq >> 8,
                hl[(size_t)t * 8 + 3], hl[(size_t)t * 8], hl[(size_t)t * 8 + 1], hl[(size_t)t * 8 + 2], hl[(size_t)t * 8 + 4], hl[(size_t)t * 8 + 5], hl[(size_t)t * 8 + 6], hl[(size_t)t * 8 + 7]);
      fclose(fp);
    }
  }
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}

}  // namespace okkt
