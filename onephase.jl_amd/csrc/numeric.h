// Device-side plan and launchers of the numeric multifrontal LDL^T (HIP, gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "symbolic.h"

namespace okkt {

// one extend-add item of a big front's column (numeric.hip: k_big_assemble*): the child's contribution-block column in the arena, its rel
// list, the rows of its contribution block, the local column, its chunk-boundary table
struct EaRec { int64_t src; int64_t rel; int rc; int jj; int64_t cut; };
static_assert(sizeof(EaRec) == 32, "two 16-byte loads per record");

// Everything the kernels need, resident in HBM for the life of a pattern.
struct DevPlan {
  int n = 0, nsuper = 0;
  int has_dup = 0;
  // symbolic structure
  int* sn_col0 = nullptr;
  int64_t* row_ptr = nullptr;
  int* rows = nullptr;
  int64_t* front_pos = nullptr;
  // Contribution blocks released (round 6): front_pos[s] is the base of the front's L PANEL (f rows x k columns, leading dimension f);
  // column c >= k of the front lives at front_pos[s] + c * f + cb_shift[s] -- in a region that fronts with disjoint lifetimes share
  // (numeric_setup: a block lives from its front's level to its parent's).  All zeros when the release is off (partitioned plans): the
  // front is then the dense f x f buffer of rounds 1 - 5.
  int64_t* cb_shift = nullptr;
  int64_t* child_ptr = nullptr;
  int* children = nullptr;
  int64_t* rel_ptr = nullptr;
  int* rel = nullptr;
  int64_t* cv_pos = nullptr;
  int64_t* aent_ptr = nullptr;
  int64_t* aent_src = nullptr;
  int* aent_dst = nullptr;
  int* perm = nullptr;
  int* sched = nullptr;  // supernode ids grouped by (level, class); for the small classes: the ROOT of a task
  int* task_lo = nullptr;  // [supernode] first supernode of the task rooted there (small fronts): the workgroup runs task_lo[s] .. s in order
  int* unit_parent = nullptr;  // [supernode] for the root of a task: root of the task that holds its parent front, -1 at a tree root
  // numeric state
  double* arena = nullptr;    // all fronts, f x f column-major each
  const double* vals = nullptr;  // caller's nzval in HBM
  double* dvals = nullptr;    // D, permuted order
  double* diagadd = nullptr;  // added to the diagonal at assembly (delta shift), permuted order
  double* xwork = nullptr;    // permuted rhs, then the solution (backward sweep)
  double* zwork = nullptr;    // z = D^-1 L^-1 P b of the forward sweep: never written over the rhs that other workgroups still gather
  double* cv = nullptr;       // solve contribution vectors
  double* wbuf = nullptr;     // W = L21*D panels of the big fronts
  int64_t* wbuf_pos = nullptr;  // [nsuper] offset of each big front's W panel, -1 for small fronts
  // big fronts only: per front-column inverted extend-add lists, inverse diagonal blocks, solve vectors
  int64_t* bigcol_base = nullptr;  // [nsuper] first global big-column index of the front, -1 for small
  int64_t* ea_ptr = nullptr;       // [n_bigcols + 1] contributions (child, jj) landing on a front column
  int* ea_child = nullptr;
  int* ea_jj = nullptr;
  int64_t* ea_pos = nullptr;     // per item: position of the child's entry in the contribution-vector array (cv_pos[child] + jj)
  int* ea_rc = nullptr;          // per item: rows of the child's contribution block
  int64_t* ea_src = nullptr;     // per item: arena offset of the child's contribution-block column
  int64_t* ea_rel = nullptr;     // per item: start of the child's rel list
  int64_t* ea_cut = nullptr;     // per item: start of the child's chunk-boundary table in cutv
  const struct EaRec* ea_rec = nullptr;   // per item: the five fields above in one 32-byte record (two 16-byte loads instead of five scattered ones)
  int* cutv = nullptr;           // per child of a big front: positions of the parent's 1024-row boundaries in its rel list
  int64_t* acol_lo = nullptr;    // [n_bigcols + 1] first A entry of each big-front column
  // solves (solve.hip): explicit inverses of the kSolveBlock-column diagonal blocks of the fronts with more than NB pivot
  // columns, a scratch copy of the same shape for the recursive inversion, partial vectors of the block products
  double* xinv = nullptr;
  double* xtmp = nullptr;
  int64_t* xinv_pos = nullptr;   // [nsuper] offset into xinv / xtmp, -1 for fronts with k <= NB (they use invl) and small fronts
  double* ypart = nullptr;       // [kMaxRhs][4][kSolveBlock] per wide front
  int64_t* ypart_pos = nullptr;
  int solve_mid = 0;             // fronts of NB + 1 .. solve_mid (<= kSolveBlock) pivot columns: block substitution in 64-column steps (k_fwd_mid / k_bwd_mid) instead of the product with the block's explicit inverse; 0 = none
  double* ythin = nullptr;       // [kMaxRhs][128] per thin front: y = X w_K between the two forward launches of a level
  int64_t* ythin_pos = nullptr;
  int* ssched = nullptr;         // big fronts per level, thin (k <= NB) then wide
  unsigned long long* sver = nullptr;   // flow sweeps of the wide fronts: per 32-row (forward) / 64-column (backward) tile, (epoch << 12) + blocks applied
  int64_t* sver_pos = nullptr;          // [nsuper] first tile word of a wide front, -1 otherwise
  int64_t xw_stride = 0, cv_stride = 0;   // distance between the right-hand sides of a batch in xwork / cv
  double* invl = nullptr;          // inverse of the unit-lower diagonal blocks, NB x NB each
  int64_t* invl_pos = nullptr;     // [nsuper]
  int* sn_owner = nullptr;     // multi-GPU partition: owner part of each supernode (-1 = top)
  int* col_owner = nullptr;    // the same per permuted column
  int* top_cols = nullptr;     // the permuted columns of the top (owner -1), ascending: what the separator-solution broadcast carries
  int* bnd = nullptr;          // boundary fronts (subtree roots under a top node)
  int64_t* bnd_cb = nullptr;   // their offsets in the contribution-block exchange buffer
  int64_t* bnd_cv = nullptr;   // ... and in the contribution-vector exchange buffer
  // pos, neg, zero, nonfinite per SLOT (kCountSlots slots of kCountStride words, one cache line each): the small-front
  // kernels spread their atomic adds over slots 1.. (60 000 one-wave workgroups adding to the same four words cost
  // 0.6 of 0.74 ms at S-metric), the big-front kernels use slot 0; slot 0 word [4] = stop flag.  Readers sum the slots.
  unsigned long long* counters = nullptr;
  long long want_pos = -1, want_neg = -1;  // >= 0: the kernels raise / obey the stop flag (retries of the delta loop), -1: off
  double* zero_page = nullptr;  // 2 KiB of zeros (source of out-of-panel LDS-DMA rows)
  // dataflow factorisation of the big fronts (dataflow.hip): per tile (i, j) of a big front's 128-block grid the number of tasks that
  // have been applied to it (updates, then the factorisation of the tile itself), TB x TB ints per front
  int df_early_pub = 1;              // multi-tile update tasks: a row tile is published as soon as its stores have drained (dataflow.hip, df_syrk_tiles)
  int df_dbg_half = 0;               // timing experiment: bulk update tasks skip the products of every other operand chunk (wrong numbers)
  int df_chain = 0;                  // chained update tasks: least distance (block columns) of a chained tile from its group's last panel; 0 = off (dataflow.hip, df_syrk_chain; OKKT_DF_CHAIN)
  int df_macro = 1;                  // update tasks on pairs of row tiles run as one macro tile (dataflow.hip, df_syrk_macro; OKKT_DF_MACRO=0: tile after tile)
  int* df_state = nullptr;
  int64_t* df_state_pos = nullptr;   // [nsuper] offset of the front's states, -1 for small fronts
};

// one task of the dataflow factorisation (dataflow_sched.cpp builds the queues, dataflow.hip runs them)
enum { kDfD = 0, kDfT = 1, kDfU = 2, kDfTU = 3, kDfTA = 4, kDfTL = 5 };
struct DfTask { int front; int type_nq; int ij; int q0; };     // type | nq << 8 | rows << 16 (update tasks: tiles (i .. i + rows - 1, j)), i | j << 16
struct DfFront { int s, f, k; };
// queue of the fronts of one level in the start order of a simulated list schedule on `workers` workers; `group` panels per
// update task where the tile allows it; model_us = the simulated makespan
void df_build_queue(const std::vector<DfFront>& fronts, int workers, int group, int rows_per_task, bool fuse_d, bool split_tu, std::vector<DfTask>& out, double* model_us,
                    bool fuse_tl = false, bool lockstep = false, bool multi_rows = true);

constexpr int kDfHeadStride = 16;   // queue counter of a level (word 0; one cache line per level)
constexpr int kCountSlots = 64, kCountStride = 16;
#ifndef OKKT_SOLVE_BLOCK
#define OKKT_SOLVE_BLOCK 1024
#endif
// columns of an explicitly inverted diagonal block (solve.hip).  Round 2 (AMD tree, root of 16 641 columns): 2048 was the fastest
// (512 / 1024 / 2048: 2.9 / 2.4 / 2.1 ms per S-metric solve).  Round 3 (dissected tree, root of 8 586): 1024 costs the S-metric solve
// nothing (1.68 against 1.60-1.72 ms), saves a doubling level of the inversion (factor 22.1 -> 21.5 ms) and a quarter of the S-C3
// solve (0.87 -> 0.63 ms); the forward error is the same (scripts/forward_error.py).  512 is NOT supported by the block-product
// kernels any more (wrong results, found in round 3): the assertion keeps the knob honest.
constexpr int kSolveBlock = OKKT_SOLVE_BLOCK;
static_assert(kSolveBlock == 1024 || kSolveBlock == 2048, "solve.hip is validated for 1024- and 2048-column inverse blocks only");
constexpr int kMaxRhs = 4;                      // right-hand sides carried through one pass over L
// sums the slots: out[0..3] = pos, neg, zero, nonfinite, out[4] = stop flag, out[5] = time-out word of the in-launch waits (synchronises `stream`)
std::string numeric_read_counts(struct Numeric& N, hipStream_t stream, unsigned long long out[6]);

// front classes by order f: 0: f<=32 (one wave), 1: f<=64, 2: f<=small_max (256 threads, LDS), 3: big
constexpr int kNumClasses = 4;
struct Segment {
  int off = 0, cnt = 0, maxf = 0, maxk = 0, minf = 1 << 30, mink = 1 << 30;
  int64_t df_off = -1;     // class 3 (big fronts): the level's task queue in Numeric::df_tasks, its length, its head counters (kDfHeadStride words), its flops
  int df_cnt = 0, df_head = -1;
  double df_flops = 0;
};
struct LevelSchedule { Segment seg[kNumClasses]; };

// big fronts of one level for the solves: thin (k <= NB: one fused forward launch) and wide (block products with X_b)
struct SolveLevel {
  int thin_off = 0, thin_cnt = 0, thin_maxf = 0, thin_maxk = 0, thin_maxr = 0;
  int wide_off = 0, wide_cnt = 0, wide_maxf = 0, wide_maxk = 0, wide_mink = 1 << 30;
};

struct LaneStreams { hipStream_t main = nullptr, masked = nullptr, panel = nullptr, aux = nullptr; };

struct Numeric {
  DevPlan d;
  int solve_fuse = 1;                    // thin fronts: the two dependent launches of a level and sweep fused into one (in-launch hand-offs); OKKT_SOLVE_FUSE=0 switches back
  int solve_flow = 0;                    // experiment (OKKT_SOLVE_FLOW=1): the forward sweep over the wide fronts of a level as ONE launch, block products and panel GEMVs handing their vectors on through tile states.  Correct, slower: 650 us instead of 534 per S-metric solve
  int solve_fuse_wide_max = 0;           // wide fronts: fused while the partial block products of a launch are at most this many workgroups (OKKT_SOLVE_FUSE_WIDE_MAX; 0 = never: measured neutral to slower, the consumers cannot start their panel loads before the hand-off)
  int* solve_flags = nullptr;            // [nsuper] monotonic y flags of the fused forward launches
  int* solve_counters = nullptr;         // [nsuper] arrival counters of the fused backward launches
  int solve_epoch = 0;
  unsigned long long* solve_counters64 = nullptr;   // [nsuper] epoch-based arrival counters of the fused wide-front launches
  unsigned long long solve_epoch64 = 0;
  int solve_split_small = 400;           // panel GEMVs of the wide fronts: launches of fewer 64-row / 64-column workgroups than this use 32 rows / 16 columns per workgroup (OKKT_SOLVE_SPLIT_SMALL; 0 = never)
  std::vector<SolveLevel> slevels, slevels_top;
  hipEvent_t inv_event = nullptr;        // recorded behind the block inversions that the factorisation started on the auxiliary stream
  bool inv_wait = false;                 // ... which the next solve has to wait for
  hipStream_t inv_stream = nullptr;      // the stream those inversions were put on
  // one event per level of `levels` behind that level's inversions: the forward sweep waits for a level's inverses when it gets
  // there, not for the root's before its first launch (the root's 0.7 ms of inversions hide behind the lower levels of the sweep)
  std::vector<hipEvent_t> inv_level_events;
  std::vector<char> inv_level_pending;
  std::vector<LevelSchedule> levels;      // subtrees owned by this part (everything when unpartitioned)
  std::vector<LevelSchedule> levels_top;  // top of the tree (part 0 of a partitioned plan only)
  int part_id = 0;
  int n_boundary = 0;
  int n_top_cols = 0;                    // columns of the top of a partitioned plan
  std::vector<void*> allocations;
  int nb = 64;
  int group = 2;   // block columns per super-step: the trailing update runs with K = group * nb
  int group_big = 4, group_big_minf = 8192;   // ... and for fronts of at least group_big_minf rows
  int group_switch_rows = 9000;               // ... until fewer rows than this are left below the super-step
  int group_one_rows = 4000;                  // single block columns once fewer rows than this are left
  int small_max = 128;
  int64_t n_small = 0, n_big = 0;
  // early exit of a factorisation whose inertia is already wrong (delta loop): checked once, before level early_level
  int early_level = -1;                  // first level of the part of the tree that holds >= 30 % of the flops (-1: none)
  bool early_check = false, early_exited = false;
  bool early_device = false;             // this factorisation is a retry: kernels stop on the device flag as well
  int64_t early_n = 0, early_m = 0;      // the inertia the caller wants
  int64_t n_tasks = 0;     // workgroup tasks of small fronts (subtrees run by one workgroup)
  int max_task_len = 0;
  hipStream_t stream = nullptr;
  // look-ahead: panels of the next super-step are factored on stream_panel while the trailing update runs
  hipStream_t stream_masked = nullptr;  // CU-masked twin of `stream`: levels that use the look-ahead fork onto it and join back
  hipStream_t stream_panel = nullptr;
  hipStream_t stream_aux = nullptr;     // off-critical-path part of the in-group panel updates
  int split_head = 1;
  int fuse_diag_trsm = 3;                // k_diag_trsm_fused, the diagonal block and the rows below it in one launch.  OKKT_FUSE_DIAG_TRSM: 0 never; 1 always (S-metric 23.7 -> 24.1 ms: the waiting trsm workgroups hold CUs the trailing update wants); 2 wherever the panels run in order (S-C5 +5 %: with many fronts per level they hold the CUs of the other fronts' diagonal blocks); 3 (default) in order AND at most OKKT_FUSE_MAX_FRONTS (8) fronts in the level: S-C3 3.53 -> 3.47 ms, S-C5 4.70 -> 4.67, S-metric unchanged
  int* chain_flags = nullptr;            // [nsuper] monotonic flags of those launches
  int chain_epoch = 0;
  // several levels of small-front tasks in ONE launch (in-launch hand-offs between a task and its children tasks): the leading
  // flow_levels levels of `levels` hold no big front; flow_off / flow_cnt is their range in `sched`.  OKKT_FLOW=0 switches back
  // to one launch per level.
  int flow = 1;
  int flow_levels = 0, flow_off = 0, flow_cnt = 0, flow_maxf = 0, flow_maxk = 0;
  int* flow_flags = nullptr;             // [3][nsuper] monotonic flags: factorisation, forward sweep, backward sweep
  int flow_epoch = 0;
  // big fronts of a level as ONE persistent launch (dataflow.hip): tasks on 128 x 128 tiles popped from a queue in a precomputed order,
  // hand-offs through per-tile states.  OKKT_DATAFLOW=0 switches back to the per-step launches of factor_sched.
  int dataflow = 1;
  int df_group = 4;                    // panels per update task (K = 128 * group) where the column allows it (OKKT_DF_GROUP; S-metric 19.1 / 18.3 / 18.2 ms at 2 / 3 / 4)
  int df_fuse_d = 1;                   // D(q + 1) in the task of TU(q): the diagonal tile passes through LDS (OKKT_DF_FUSE_D=0: a task of its own)
  int df_split_tu = 1;                 // block rows of more than 64 rows: TU(q) as two tasks on two workers, TA(q) (upper 64 rows) and TU(q) (OKKT_DF_SPLIT_TU=0: one)
  std::vector<int64_t> front_pos_host;     // the device plan's front_pos (panel bases)
  int64_t arena_doubles = 0;               // doubles of the front arena as allocated (panels + the shared contribution-block region)
  int64_t cb_region_doubles = 0;           // ... of which the contribution-block region
  int release_cb = 1;                      // OKKT_RELEASE_CB (0: every front keeps its f x f buffer for the plan's lifetime)
  int df_lockstep = 0;                 // TU(q) in lockstep with the 32-column blocks of D(q) (df_tu_lock; OKKT_DF_LOCKSTEP=0: the split TA / TU of rounds 4 - 5 behind the whole of D(q))
  int df_fuse_tl = 1;                  // T(i, q) with the last update of its tile inside the task (TL), q >= 1 (OKKT_DF_FUSE_TL=0: separate tasks)
  int df_rows = 1;                     // row tiles per bulk update task (OKKT_DF_ROWS; 2 and 4 measured slower: the coarser tasks cost the schedule more than the shared prologue saves)
  int df_workers = 256;                // workers of the simulated schedule (and the grid of the launch): one workgroup per CU
  DfTask* df_tasks = nullptr;
  int* df_heads = nullptr;
  int n_df_heads = 0;
  int64_t df_state_ints = 0;
  int diag2 = 1;                         // k_big_diag2 (role-split, pipelined) instead of k_big_diag; OKKT_DIAG2=0 switches back
  int lookahead = 1;
  int sb_tail_rows = 4000;               // the inversion of the finished diagonal blocks starts once fewer rows than this remain
  int la_min_tiles = 600;                // rest triangle must hold at least this many 128 x 128 tiles
  std::vector<hipEvent_t> la_events;
  size_t la_used = 0;
  double* vals_owned = nullptr;  // staging buffer for host-side nzval
  int64_t nnz_in = 0;
  // optional per-launch timing of the dominant kernel (k_big_syrk) with HIP events on N.stream
  bool profile = false;
  std::vector<hipEvent_t> prof_events;   // pairs (start, stop)
  size_t prof_used = 0;                  // events consumed since the last reset
  std::vector<double> prof_flops;        // algorithmic flops of each profiled launch
  std::vector<std::pair<int, int>> big_fk;  // (f, k) of sched entries (host copy, big fronts only need it)
  std::vector<int> sched_host;
  std::vector<int> sn_f, sn_k;           // per supernode
};

// returns "" or an error message
std::string numeric_setup(const Symbolic& S, const SymbolicOptions& opts, hipStream_t stream, Numeric& N);
void numeric_release(Numeric& N);

// enqueue the whole numeric factorisation on N.stream (no sync); values read from d_vals
// which = 0: the local schedule, 1: the top-of-tree schedule
std::string numeric_factor_enqueue(Numeric& N, const double* d_vals, double tol, int which = 0, bool reset_counters = true);
// forward (L, D^-1 fused) / backward (L') sweep over one schedule for R = 1, 2 or 4 right-hand sides held in xwork
std::string solve_fwd_enqueue(Numeric& N, int which, int R);
std::string solve_bwd_enqueue(Numeric& N, int which, int R);
std::string solve_setup(const Symbolic& S, Numeric& N);
std::string solve_invert_enqueue(Numeric& N, hipStream_t st, const SolveLevel& L, int b_lo, int b_hi);
// xwork[q][k] = rhs[q * stride + perm[k]] for q < nr (zero for nr <= q < R)  /  sol[q * stride + perm[k]] (+)= xwork[q][k]
void solve_permute_in(const Numeric& N, const double* d_rhs, int64_t stride, int nr, int R);
void solve_permute_out(const Numeric& N, double* d_sol, int64_t stride, int nr, int R, bool accumulate);
// what = 0 contribution blocks, 1 contribution vectors; unpack = 0: mine -> buffer, 1: buffer -> the others' slots
std::string numeric_dist_pack(Numeric& N, int what, int unpack, double* d_buf);
// mode 0: the top's columns of xwork -> buf, packed (n_top_cols doubles); 1: buf -> xwork on the top's columns; 2: owned part of the solution,
// original order -> buf (n doubles, zeros elsewhere)
std::string numeric_dist_x(Numeric& N, int mode, double* d_buf);
// out[0..3] = pos, neg, zero, nonfinite summed over the counter slots, on the device (no synchronisation)
void numeric_sum_counts_device(Numeric& N, long long* d_out4);
// enqueue forward/diagonal/backward solves for the R right-hand sides already stored (permuted) in d.xwork
std::string numeric_solve_enqueue(Numeric& N, int R);
// diagadd[iperm] = (orig index < nshift) ? delta : 0, via perm
void launch_set_shift(const Numeric& N, double delta, int64_t nshift);
// dataflow.hip: the big fronts of one level (class-3 segment g, already assembled) as one persistent launch on `st`
std::string df_setup(Numeric& N);
const char* df_build_flags();      // which optional roles the dataflow kernel of this library carries (dataflow.hip)
std::string df_launch(Numeric& N, const DevPlan& P, const Segment& g, hipStream_t st, double tol);

}  // namespace okkt
