/*
 * okkt.h -- C ABI of libonephase_kkt.so: the MI355X-native KKT linear-system path of the
 * one-phase interior point method (reference behaviour: ohinder/OnePhase.jl).
 *
 * Two drop-in levels:
 *
 *  (1) linear-solver level -- what a `linear_solver_HIP <: abstract_linear_system_solver`
 *      binds in place of `linear_solver_JULIA` (CHOLMOD):
 *        initialize!  src/linear_system_solvers/linear_system_solvers.jl:40   -> okkt_create
 *        ls_factor!   src/linear_system_solvers/julia.jl:21-97                -> okkt_analyze + okkt_factor
 *        ls_solve!    src/linear_system_solvers/julia.jl:99-103               -> okkt_solve
 *        ls_solve     src/linear_system_solvers/julia.jl:105-113              -> okkt_solve
 *        finalize!    src/linear_system_solvers/linear_system_solvers.jl:44   -> okkt_destroy
 *        inertia_status  src/linear_system_solvers/linear_system_solvers.jl:48-91 -> okkt_inertia + return code
 *
 *  (2) KKT-system level -- what a `HIP_KKT_solver <: abstract_KKT_system_solver` binds in
 *      place of Schur_KKT_solver / Symmetric_KKT_solver, keeping everything device-resident:
 *        form_system!                      src/kkt_system_solver/schur.jl:47-62, symmetric.jl:35-53 -> okkt_kkt_form_system
 *        update_delta_vecs! + factor!      schur.jl:64-87, symmetric.jl:55-57,85-102, kkt_system_solver.jl:98-113,190-204 -> okkt_kkt_factor
 *        compute_direction_implementation! schur.jl:89-182, symmetric.jl:59-83 (+ update_kkt_error! kkt_system_solver.jl:67-96) -> okkt_kkt_compute_direction
 *        ipopt_strategy!                   src/IPM/delta_strategy.jl:37-114 -> okkt_kkt_ipopt_strategy
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the boundary.  Functions
 * return OKKT_OK (0) or a negative okkt_status; the factor calls return 1 (inertia correct),
 * 0 (inertia wrong / zero or non-finite pivot) or a negative error -- the same 1/0 contract as
 * ls_factor!.  All calls are blocking (the handle's HIP stream is synchronised before return).
 * A handle is not thread-safe; several handles may coexist.
 *
 * There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 * OKKT_ERR_NO_DEVICE (okkt_create succeeds only with opts.host_symbolic_only = 1, which permits
 * okkt_analyze and the query functions and nothing else).
 */
#ifndef OKKT_H
#define OKKT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct okkt_solver_s* okkt_handle;
typedef struct okkt_kkt_s* okkt_kkt_handle;

typedef enum {
  OKKT_OK = 0,
  OKKT_ERR_INVALID = -1,     /* bad argument / call out of order */
  OKKT_ERR_NO_DEVICE = -2,   /* no HIP device, or handle is host_symbolic_only */
  OKKT_ERR_HIP = -3,         /* a HIP runtime call failed (see okkt_last_error) */
  OKKT_ERR_ALLOC = -4,
  OKKT_ERR_INTERNAL = -5
} okkt_status;

/* sym_kind: the `sym` symbol of the reference's solver constructors (julia.jl:11) */
#define OKKT_SYM_DEFINITE 0   /* :definite  -- Cholesky semantics: success <=> every pivot > 0     */
#define OKKT_SYM_SYMMETRIC 1  /* :symmetric -- LDL^T, inertia from sign(D) with tolerance 1e-20   */

/* kkt_kind: pars.kkt.kkt_solver_type (parameters.jl:30) */
#define OKKT_KKT_SCHUR 0      /* Q = H + J' diag(y/s) J       (n x n),     Cholesky semantics */
#define OKKT_KKT_SYMMETRIC 1  /* K = [[H J'];[J -diag(s/y)]]  (n+m square), LDL^T inertia (n,m,0) */
#define OKKT_KKT_CLEVER_SYMMETRIC 2  /* parallel rows of J merged first: M = [[H 0];[J_new -U_new]] (n+m_new square),
                                      * inertia (n, m_new, 0); Clever_Symmetric_KKT_solver, clever_symmetric.jl:25-519 */
#define OKKT_KKT_SCHUR_DIRECT 3      /* Schur_KKT_solver_direct (schur_direct.jl:3-66, kkt_system_solver.jl:270-276): the Schur system of the
                                      * FACTORISED iterate, but rhs terms, dy and ds = (comp_r - dy .* s) ./ y from the CURRENT iterate
                                      * (the one of the last okkt_kkt_system_rhs) */
/* kkt_system_rescale of the clever-symmetric system (parameters.jl:24-28, clever_symmetric.jl:307-325) */
#define OKKT_RESCALE_NONE 0
#define OKKT_RESCALE_U_ONLY 1
#define OKKT_RESCALE_U_AND_X 2

typedef struct {
  int32_t device;             /* HIP device ordinal; -1 = current device */
  int32_t host_symbolic_only; /* 1: never touch the GPU (analysis + queries only; CPU-side tests) */
  int32_t ordering;           /* 0 automatic (default): AMD, replaced by level-structure nested dissection when the AMD
                                 elimination tree is a path of small fronts (banded KKT systems: a dependent pivot chain on a
                                 GPU), and by multilevel nested dissection when that needs at least 10 % fewer factor flops (n >= 10 000; both candidates are
                                 computed side by side); 1 natural, 2 user permutation (okkt_set_perm), 3 AMD always, 4 level-structure nested
                                 dissection always, 5 multilevel nested dissection always */
  int32_t relax_always;       /* supernode amalgamation knobs, <=0 = default */
  int32_t relax_small;
  int32_t relax_mid;
  double relax_small_frac;
  double relax_mid_frac;
  double relax_any_frac;
  double inertia_tol;         /* |d| <= tol counts as a zero pivot (julia.jl:73); default 1e-20 */
  int32_t small_front_max;    /* fronts of order <= this use the LDS-resident kernel; <=0 default */
  int32_t panel_nb;           /* block-column width in the big-front kernels; <=0 default */
  int32_t early_exit;         /* 1: a factorisation may stop once its inertia is decided wrong (see okkt_set_early_exit); default 0 */
  int32_t reserved;
} okkt_opts;

typedef struct {
  int64_t pos, neg, zero, nonfinite; /* counts over diag(D); julia.jl:72-78 */
} okkt_inertia;

typedef struct {
  int64_t n;              /* order of the analysed matrix */
  int64_t nnz_lower;      /* input entries with row >= col */
  int64_t nnzL;           /* sum_j c_j (structural, no relaxation zeros) */
  int64_t nnzL_stored;    /* panel entries actually stored (with relaxation zeros) */
  double flops_exact;     /* sum_j c_j^2  (SURVEY 8d factor flops) */
  double flops_stored;    /* dense-front flops executed */
  int64_t arena_bytes;    /* HBM bytes of the front arena as allocated (before the device plan exists: sum f^2 * 8) */
  int64_t nsuper;
  int64_t nlevels;
  int64_t max_front;
  int64_t n_small_fronts, n_big_fronts;
  int64_t sum_rowidx;     /* sum_s f_s (row-index entries) */
  double analyze_seconds; /* host time of the last analysis */
  double last_factor_ms;  /* device time of the last numeric factorisation (hipEvent) */
  double last_solve_ms;   /* device time of the last solve */
  uint64_t pattern_hash;
  int64_t n_analyze_calls; /* how many times a new pattern forced a re-analysis */
  int64_t ordering_used;   /* 0 AMD, 1 natural, 2 user, 4 level-structure nested dissection, 5 multilevel nested dissection */
  int64_t critical_pivots; /* pivots on the longest leaf-to-root path of the supernodal elimination tree */
  int64_t top_separator;   /* multilevel dissection: vertices of the top-level separator (-1: none) */
  int64_t amd_skipped;     /* automatic ordering: 1 = minimum degree was abandoned (small top separator), no flop comparison was made */
  double flops_other;      /* automatic ordering: factor flops of the candidate that lost the comparison (0: none, or skipped) */
  int64_t arena_dense_bytes; /* what the front arena would take with a dense f x f buffer per front (sum f^2 * 8: the layout of rounds 1 - 5);
                              * arena_bytes is the arena as allocated -- L panels + the region the contribution blocks share by lifetime */
} okkt_stats;

/* ---- level 1: linear solver ------------------------------------------------------------ */
int okkt_default_opts(okkt_opts* opts);
int okkt_create(okkt_handle* out, const okkt_opts* opts /* NULL = defaults */);
int okkt_destroy(okkt_handle h);
const char* okkt_last_error(okkt_handle h);
const char* okkt_version(void);

/* user permutation for opts.ordering == 2: perm[new] = old, 0-based; call before okkt_analyze */
int okkt_set_perm(okkt_handle h, const int64_t* perm, int64_t n);
/* pattern of a square CSC matrix (only row >= col is used; upper entries are ignored, as under
 * Symmetric(A,:L), julia.jl:34,52).  Cached by pattern (exact comparison with the analysed colptr/rowval): re-calling with the same pattern
 * is free, so the reference-shaped ls_factor!(A,...) can call it every time. */
int okkt_analyze(okkt_handle h, int64_t dim, const int64_t* colptr, const int64_t* rowval, int index_base);
int okkt_get_perm(okkt_handle h, int64_t* perm_out /* [dim], perm[new]=old, 0-based */);
int okkt_get_stats(okkt_handle h, okkt_stats* out);
int okkt_get_etree(okkt_handle h, int64_t* parent_out /* [dim] */, int64_t* colcount_out /* [dim] */);

/* numeric factorisation of the analysed pattern with values nzval (same order as rowval).
 * n + m must equal dim.  Returns 1 / 0 / <0 like ls_factor! (julia.jl:21-97). */
int okkt_factor(okkt_handle h, const double* nzval, int64_t n, int64_t m, int sym_kind, okkt_inertia* inertia_out);
/* same, nzval already resident in HBM (device pointer) */
int okkt_factor_dev(okkt_handle h, const double* d_nzval, int64_t n, int64_t m, int sym_kind, okkt_inertia* inertia_out);
/* sol = F \ rhs for nrhs right-hand sides stored one after another (julia.jl:101,110).  rhs may alias sol. */
int okkt_solve(okkt_handle h, const double* rhs, double* sol, int64_t nrhs);
int okkt_solve_dev(okkt_handle h, const double* d_rhs, double* d_sol, int64_t nrhs);
/* diag(F): the D of LDL^T in pivot (permuted) order, as `diag(solver._factor)` (julia.jl:72) */
int okkt_get_diag(okkt_handle h, double* d_out /* [dim] */);
/* L as CSC in permuted numbering (unit diagonal not stored), for parity tests; pass NULLs to size */
int okkt_get_factor_csc(okkt_handle h, int64_t* colptr_out, int64_t* rowval_out, double* val_out, int64_t* nnz_out);

/* device-memory helpers so that callers without a HIP binding (ctypes, Julia) can keep inputs in HBM */
int okkt_dev_alloc(okkt_handle h, int64_t bytes, void** d_ptr_out);
int okkt_dev_free(okkt_handle h, void* d_ptr);
int okkt_dev_upload(okkt_handle h, void* d_dst, const void* src, int64_t bytes);
int okkt_dev_download(okkt_handle h, void* dst, const void* d_src, int64_t bytes);
/* ls_factor! only returns the inertia flag and the reference never solves with a factorisation that failed it
 * inside the delta loop (it updates delta and refactors, delta_strategy.jl:37-114).  With early exit enabled okkt_factor /
 * okkt_factor_dev stop before the top of the elimination tree when the pivots counted so far already decide a wrong
 * inertia; they return 0, `out` holds the counts of the columns eliminated so far, and okkt_solve is refused until the
 * next complete factorisation.  Off by default: outside the delta loop the reference DOES solve with a factorisation whose
 * flag was 0 (the refactorisation after a failed step, one_phase.jl:241), so only a caller that discards failed factors
 * may turn it on (the KKT level does so inside okkt_kkt_ipopt_strategy / okkt_kkt_factor_trial only). */
int okkt_set_early_exit(okkt_handle h, int enable);
/* the handle's HIP stream (hipStream_t as void*), for callers that time with their own events */
void* okkt_get_stream(okkt_handle h);
/* per-launch timing of the dominant kernel (k_front_dataflow, the persistent launch that factors the big fronts of one level;
 * with OKKT_DATAFLOW=0 the FP64-MFMA trailing update k_big_syrk of the per-step schedule): HIP events are recorded around
 * every launch on the handle's stream while enabled; okkt_get_profile returns the number of launches since enabling, their
 * summed duration and their summed algorithmic flops (k f^2 - k^2 f + k^3 / 3 per front of a dataflow launch; rem * (rem + 1) * nb
 * per front and block column of a trailing update, DESIGN.md "Kernels") */
int okkt_profile_dominant(okkt_handle h, int enable);
int okkt_get_profile(okkt_handle h, int64_t* n_launches, double* total_ms, double* total_flops);
/* Test hook, host only (no device, no handle): the task queue of the dataflow launch (csrc/dataflow.hip) for one level of
 * nfronts big fronts of orders f[] with k[] pivot columns each, as it would be uploaded for `workers` workers, `group & 255` panels
 * and `(group >> 8) & 255` row tiles (0 = 1) per bulk update task; bit 16 of `group`: D(q + 1) rides in TU(q) (bit 1 of nq in its task); bit 17: a block row of more than 64 rows is split between TA(q) (type 4, its upper 64 rows) right before TU(q) (bit 2 of nq); bit 18: the panel tiles from panel 1 on carry the last update of their tile (type 5 = TL(i, q): U(i, q, q - 1, 1) then T(i, q); that update is not a task of its own).  tasks receives 4 ints per task: front index, type | nq << 8 | rows << 16 (type 0 = D diagonal tile, 1 = T panel tile,
 * 2 = U update of the tiles (i .. i + rows - 1, j), 3 = TU: panel tile (i, j) and the diagonal tile (i, i), i = j + 1), i | j << 16 (tile row / column; for T: j = the panel), q0 (first panel of an update).  Returns the number of
 * tasks (also when it exceeds cap; only cap tasks are written), or a negative error code.  tests/test_dataflow_queue.py replays
 * the queue on a dense matrix with numpy and checks the dependency order. */
int64_t okkt_debug_dataflow_queue(int32_t nfronts, const int32_t* f, const int32_t* k, int32_t workers, int32_t group,
                                  int32_t* tasks, int64_t cap, double* model_us);


/* ---- multi-GPU: subtree-to-GPU sharding of ONE factorisation (one process per GPU) -------------------
 * No reference counterpart (the reference is single-process).  Every rank analyses the same
 * pattern and calls okkt_dist_set_partition(nparts, its part id): disjoint elimination-tree subtrees are
 * assigned to the parts (flop-balanced, deterministic), the ancestors of the cut ("top") to part 0.  The
 * caller moves three flat device buffers between ranks with its own collective (RCCL reduce / broadcast):
 *   factor: okkt_dist_factor_local -> okkt_dist_cb(buf, 0) [reduce(sum) to part 0] okkt_dist_cb(buf, 1) ->
 *           okkt_dist_factor_top (part 0) -> okkt_dist_counts [all-reduce(sum)] -> okkt_dist_finish -> 1/0
 *   solve : okkt_dist_solve_begin(rhs) -> okkt_dist_cv(buf, 0) [reduce(sum) to part 0] okkt_dist_cv(buf, 1) ->
 *           okkt_dist_solve_top (part 0) -> okkt_dist_x(buf, 0) on part 0 [broadcast] okkt_dist_x(buf, 1) ->
 *           okkt_dist_solve_end -> okkt_dist_x(sol, 2) [reduce(sum)]: the solution in original order.
 * Buffers must be zero before the pack calls (every slot has exactly one writer, the sum is exact). */
int okkt_dist_set_partition(okkt_handle h, int nparts, int part_id);
int okkt_dist_info(okkt_handle h, int64_t* cb_doubles, int64_t* cv_doubles, int64_t* n_boundary,
                   double* part_flops_out /* [nparts] or NULL */, double* top_flops_out);
int okkt_dist_get_owner(okkt_handle h, int64_t* sn_owner_out, int64_t* col_owner_out, int64_t* sn_parent_out);
int okkt_dist_factor_local(okkt_handle h, const double* d_nzval, int64_t n, int64_t m, int sym_kind);
int okkt_dist_cb(okkt_handle h, double* d_buf, int unpack);
int okkt_dist_factor_top(okkt_handle h);
int okkt_dist_counts(okkt_handle h, int64_t out[4]);
int okkt_dist_finish(okkt_handle h, const int64_t total[4]);
int okkt_dist_solve_begin(okkt_handle h, const double* d_rhs);
int okkt_dist_cv(okkt_handle h, double* d_buf, int unpack);
int okkt_dist_solve_top(okkt_handle h);
int okkt_dist_x(okkt_handle h, double* d_buf, int mode);
int okkt_dist_solve_end(okkt_handle h);

/* The same two sequences with the collectives INSIDE the library, on RCCL directly (ncclReduce / ncclBroadcast / ncclAllReduce
 * enqueued on the handle's stream between the kernels: no host synchronisation between the phases, one at the end) -- what a
 * Julia `linear_solver_HIP` uses when one process per GPU shares a factorisation; torch.distributed is not involved.
 * librccl is opened at run time (dlopen "librccl.so.1" / "librccl.so"; OKKT_RCCL_PATH overrides), so the library loads and the
 * single-GPU path works where RCCL is absent.  Protocol: one rank calls okkt_dist_unique_id and ships the 128 bytes to the
 * others by any means (MPI, a file, torch's store); every rank then calls okkt_dist_set_partition(nranks, rank) and
 * okkt_dist_comm_init(h, nranks, rank, id).  okkt_dist_factor returns the same 1 / 0 flag and the same summed pivot counts
 * on every rank; okkt_dist_solve leaves the whole solution (original order) in d_sol on every rank. */
int okkt_dist_unique_id(void* id_out /* 128 bytes */);
int okkt_dist_comm_init(okkt_handle h, int nranks, int rank, const void* id /* 128 bytes */);
int okkt_dist_comm_destroy(okkt_handle h);
int okkt_dist_factor(okkt_handle h, const double* d_nzval, int64_t n, int64_t m, int sym_kind, okkt_inertia* inertia_out);
int okkt_dist_solve(okkt_handle h, const double* d_rhs, double* d_sol);

/* ---- level 2: device-resident KKT system solver ----------------------------------------- */
typedef struct {
  double delta_start, delta_min, delta_max, delta_inc, delta_dec, delta_zero; /* parameters.jl:147-158 */
  int32_t ItRefine_Num;   /* parameters.jl:20 (3) -- Schur refinement rounds (first one is the plain solve) */
  int32_t max_it;         /* delta_strategy.jl:40 (500) */
} okkt_kkt_pars;

/* device times (HIP events on the handle's stream) of the phases of the last calls, milliseconds; 0 = not run yet.
 * what the reference's class_advanced_timer labels "SCHUR/form_system",
 * "SCHUR/delta_vecs", "<ls>/factorize", "KKT/rhs", "<ls>/ls_solve", "SCHUR/iterative_refinement/residual",
 * "SCHUR/kkt_err" measure on the host */
typedef struct {
  double assemble_ms;      /* okkt_kkt_form_system: kernels only (after the H, J, s, y uploads) */
  double upload_ms;        /* ... the host -> device copies of that call */
  double shift_ms;         /* okkt_kkt_factor: the delta shift (update_delta_vecs!) */
  double factor_ms;        /* ... the numeric factorisation */
  double rhs_ms;           /* okkt_kkt_system_rhs kernels */
  double solve_ms;         /* okkt_kkt_compute_direction: all triangular solves together */
  double refine_ms;        /* ... residual evaluations of the refinement rounds and the rhs / dy / ds vector work */
  double kkt_err_ms;       /* ... update_kkt_error! */
  double direction_ms;     /* ... the whole call on the device */
  int32_t n_solves;        /* triangular solves of the last direction */
  int32_t reserved;
} okkt_kkt_timers;

typedef struct {
  double error_D, error_P, error_mu, overall, rhs_norm, ratio; /* Class_kkt_error, kkt_system_solver.jl:49-65 */
} okkt_kkt_error;

int okkt_kkt_default_pars(okkt_kkt_pars* pars);
int okkt_kkt_create(okkt_kkt_handle* out, const okkt_opts* opts, int kkt_kind);
int okkt_kkt_destroy(okkt_kkt_handle k);
const char* okkt_kkt_last_error(okkt_kkt_handle k);
/* the underlying linear-solver handle (for stats / permutation queries) */
okkt_handle okkt_kkt_linear_solver(okkt_kkt_handle k);
/* structure of the iterate cache (Class_iterate.jl:4-20): H n x n CSC lower triangle only
 * (Class_cutest.jl:548), J m x n CSC.  Builds the pattern of Q or K, analyses it, builds maps. */
int okkt_kkt_set_structure(okkt_kkt_handle k, int64_t n, int64_t m,
                           const int64_t* H_colptr, const int64_t* H_rowval,
                           const int64_t* J_colptr, const int64_t* J_rowval, int index_base);
/* form_system!: values of H and J and the point (s, y); computes schur_diag on the device */
int okkt_kkt_form_system(okkt_kkt_handle k, const double* H_nzval, const double* J_nzval,
                         const double* s, const double* y);
/* diag_min(kkt_solver) (kkt_system_solver.jl:291-294) */
int okkt_kkt_diag_min(okkt_kkt_handle k, double* out);
/* factor!(kkt_solver, delta): shift the first n diagonal entries by delta, refactor; 1 / 0 / <0.  Always a complete
 * factorisation (unless the linear-solver handle was created with opts.early_exit = 1 / okkt_set_early_exit): the
 * reference computes a direction from a factor! whose inertia flag was 0 after a failed step (one_phase.jl:231-242,
 * take_step2!), so the factor must exist whatever the flag says. */
int okkt_kkt_factor(okkt_kkt_handle k, double delta, okkt_inertia* inertia_out);
/* factor! as the delta loop uses it (delta_strategy.jl:37-114): a factorisation with the wrong inertia is thrown away by the
 * caller, so it may stop as soon as the pivot counts decide a failure (returns 0; a direction cannot be computed from it --
 * okkt_kkt_compute_direction then fails with OKKT_ERR_INVALID until the next complete factorisation).  A success is always
 * a complete factorisation.  okkt_kkt_ipopt_strategy uses this internally (OKKT_EARLY_EXIT=0 in the environment disables it). */
int okkt_kkt_factor_trial(okkt_kkt_handle k, double delta, okkt_inertia* inertia_out);
int okkt_kkt_get_timers(okkt_kkt_handle k, okkt_kkt_timers* out);
/* ipopt_strategy!: returns 1 on :success, 0 on :failure (delta > delta_max), <0 on error */
int okkt_kkt_ipopt_strategy(okkt_kkt_handle k, double delta_prev, const okkt_kkt_pars* pars,
                            int32_t* num_fac_out, double* delta_out);
/* compute_direction! for nrhs reduction-factor triples in one pass over the factor: the probe of the aggressive step
 * (Reduct_affine, take_step.jl:2-3) and the candidates of take_step2! (take_step.jl:34-66) share the factorised system.
 * System_rhs (system_rhs.jl:57-73) is evaluated on the device for every triple from the iterate of the last okkt_kkt_system_rhs;
 * the triangular solves (and the Schur refinement rounds) carry up to four right-hand sides per sweep over L.
 * etas: nrhs x (eta_P, eta_D, eta_mu); dx: nrhs x n, dy, ds: nrhs x m (NULL: not downloaded); err: nrhs records or NULL.
 * Schur, Schur-direct and symmetric systems; 1 <= nrhs <= 16. */
int okkt_kkt_compute_directions(okkt_kkt_handle k, int32_t nrhs, const double* etas, int32_t ItRefine_Num,
                                double* dx, double* dy, double* ds, okkt_kkt_error* err);
/* is_diag_dom(kkt_solver.Q[1:n,1:n]) (delta_strategy.jl:1-9) at the delta of the last factor call, as a device scan:
 * *out = 1 dominant, 0 not, -1 not evaluated (clever-symmetric system).  ipopt_strategy! runs it after every failed attempt and
 * prints "WARNING: Inertia calculation incorrect" when it holds (delta_strategy.jl:94-98): okkt_kkt_ipopt_strategy does the same
 * scan and okkt_kkt_diag_dom_warnings returns how many of its failed attempts would have printed the warning. */
int okkt_kkt_is_diag_dom(okkt_kkt_handle k, int32_t* out);
int okkt_kkt_diag_dom_warnings(okkt_kkt_handle k, int32_t* count);
/* the tail of estimate_y_tilde (guess-vars.jl:155-160) with the factor and the Jacobian of the handle: y = -J (F \ (-g)) */
int okkt_kkt_estimate_y_tilde(okkt_kkt_handle k, const double* g, double* y_out);
/* System_rhs(it, reduct) (system_rhs.jl:57-73): dual_r = -(grad - J'y + eta_mu*mu*pen*J'1)(1 - eta_D),
 * primal_r = -(cons - s)(1 - eta_P), comp_r = eta_mu*mu - s.*y, at the CURRENT iterate.  J_nzval_cur = NULL
 * uses the J values of the factorised iterate (form_system) */
int okkt_kkt_system_rhs(okkt_kkt_handle k, const double* J_nzval_cur, const double* grad, const double* cons,
                        const double* s, const double* y, double mu, double a_norm_penalty,
                        double eta_P, double eta_D, double eta_mu, double* dual_r, double* primal_r, double* comp_r);
/* compute_direction!: rhs triple (dual_r[n], primal_r[m], comp_r[m]) -> (dx[n], dy[m], ds[m]) + N err.
 * dual_r = primal_r = comp_r = NULL: the rhs that the last okkt_kkt_system_rhs left on the device (no copy);
 * dx = dy = ds = NULL: the direction stays on the device only (step-side functions, okkt_kkt_get_direction).
 * OKKT_KKT_SCHUR_DIRECT reads s, y and J of the iterate of the last okkt_kkt_system_rhs (current_it). */
int okkt_kkt_compute_direction(okkt_kkt_handle k, const double* dual_r, const double* primal_r,
                               const double* comp_r, int32_t ItRefine_Num,
                               double* dx, double* dy, double* ds, okkt_kkt_error* err_out);
int okkt_kkt_get_direction(okkt_kkt_handle k, double* dx, double* dy, double* ds);
/* the assembled matrix values in the order of the analysed pattern (tests), and schur_diag */
int okkt_kkt_get_matrix(okkt_kkt_handle k, int64_t* dim_out, int64_t* nnz_out,
                        int64_t* colptr_out, int64_t* rowval_out, double* nzval_out);
int okkt_kkt_get_schur_diag(okkt_kkt_handle k, double* out /* [n] */);

/* ---- Clever_Symmetric only -------------------------------------------------------
 * initialize!(::Clever_Symmetric_KKT_solver, it) = compute_indicies(get_jac(it)) (clever_symmetric.jl:53-61,
 * 200-246): rows of J that are exact multiples of each other (same pattern, ||a_i - a_j * ratio||_2 < 1e-16)
 * are grouped on the host with the reference's ordering rule (compare_columns, :107-155); the reduced matrix
 * is analysed here.  Call once after okkt_kkt_set_structure and before the first okkt_kkt_form_system; the
 * grouping is kept for the life of the handle, as in the reference. */
int okkt_kkt_compute_indicies(okkt_kkt_handle k, const double* J_nzval, int64_t* m_new_out);
/* the grouping, 0-based: first_para_indicies [m_new] (sorted first rows = para_row_info[g].first),
 * group_ptr [m_new + 1], and per member in ls order: ind, ratio; after a form_system also u, g and the
 * group's combined u (update_indicies!, clever_symmetric.jl:262-287).  Any pointer may be NULL. */
int okkt_kkt_get_indicies(okkt_kkt_handle k, int64_t* first_para_indicies, int64_t* group_ptr, int64_t* member_ind,
                          double* member_ratio, double* member_u, double* member_g, double* group_u);
/* diag_rescale used by the next okkt_kkt_form_system: mode OKKT_RESCALE_*, mu = iter.point.mu,
 * x_norm_inf = norm(iter.point.x, Inf) (create_diag_rescale_*, clever_symmetric.jl:307-319); default NONE */
int okkt_kkt_set_rescale(okkt_kkt_handle k, int mode, double mu, double x_norm_inf);

/* ---- step-side vector kernels ---------------------------------------------------
 * The reductions and SpMVs simple_ls runs either side of a step (line_search.jl:36-199), on the state the
 * handle already holds: "iter" = the point (s, y) and J, H of the last okkt_kkt_form_system, "dir" = the
 * direction left on the device by the last okkt_kkt_compute_direction (an error if there is none).  Vector
 * arguments are host pointers of the stated length; min / max reductions propagate NaN as Julia's do and are
 * bit-exact, sums are taken in a fixed order (reproducible, not Julia's order). */
/* replace the resident direction (scale_direction of line_search.jl:10-19, corrections, tests): dx [n], dy, ds [m] */
int okkt_kkt_set_direction(okkt_kkt_handle k, const double* dx, const double* dy, const double* ds);
/* simple_max_step(iter.point.s, dir.s, lb_s_predict(iter, dir, pars)) (frac_boundary.jl:3-15,31-35;
 * line_search.jl:40-41); ex = pars.ls.fraction_to_boundary_predict_exp; also returns norm(dir.x, Inf) */
int okkt_kkt_max_step_primal(okkt_kkt_handle k, const double* frac_bd_predict /* [m] */, double ex,
                             double* step_size_P, double* dx_norm_inf);
/* all(s_new .>= lb_s(iter, dir, pars)) of move_primal (move.jl:15-17; frac_boundary.jl:22-28): 1 / 0 */
int okkt_kkt_s_bound_ok(okkt_kkt_handle k, const double* s_new /* [m] */, const double* frac_bd /* [m] */, double ex,
                        int32_t* ok);
/* lb, ub = dual_bounds(candidate, candidate.point.y, dir.y, comp_feas); ub = min(ub, simple_max_step(
 * candidate.point.y, dir.y, lb_y(iter, dir, pars))) (move.jl:28-80; frac_boundary.jl:17-20;
 * line_search.jl:84-86), with the sequential semantics of the reference's loop */
int okkt_kkt_dual_step_range(okkt_kkt_handle k, const double* s_cand /* [m] */, const double* y_cand /* [m] */,
                             double mu_cand, double comp_feas, const double* frac_bd /* [m] */, double* lb, double* ub);
/* out = { phi_predicted_reduction_primal_dual, norm(comp(iter), Inf), norm(comp_predicted(iter, dir, step), Inf),
 * merit_function_predicted_reduction } (eval.jl:11-13,117-120,236-273); grad = get_grad(iter) [n], mu =
 * iter.point.mu, dmu = dir.mu, a_norm_penalty = iter.a_norm_penalty_par */
int okkt_kkt_predicted_reduction(okkt_kkt_handle k, const double* grad, double mu, double dmu, double a_norm_penalty,
                                 double step_size, double out[4]);
/* step_size_D of move_dual (move.jl:82-118) for move_primal_seperate_to_dual: dual_ls 1 / 3 = the least-squares
 * step on [scale_D * dual residual; -scale_mu * comp] clamped to [max(lb, min(ub, step_size_P)), ub], any other
 * dual_ls = ub.  The candidate: J_nzval_cand (pattern of okkt_kkt_set_structure; NULL = the J of form_system),
 * grad_cand [n], s_cand, y_cand [m] (y not yet moved), mu_cand */
int okkt_kkt_dual_step(okkt_kkt_handle k, const double* J_nzval_cand, const double* grad_cand, const double* s_cand,
                       const double* y_cand, double mu_cand, double a_norm_penalty, double step_size_P, double lb,
                       double ub, int dual_ls, double scale_D, double scale_mu, double* step_size_D);

#ifdef __cplusplus
}
#endif
#endif /* OKKT_H */
