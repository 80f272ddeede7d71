// Host-side symbolic analysis for the supernodal multifrontal LDL^T engine.
//
// What CHOLMOD's "analyze" step does for the reference on every ls_factor! call
// (/root/reference/src/linear_system_solvers/julia.jl:34,52 -> cholesky/ldlt ->
// cholmod_analyze: AMD ordering, elimination tree, postorder, column counts,
// supernodes) is done here ONCE per sparsity pattern.  The result is a static plan
// (front layout in HBM, scatter maps, extend-add index lists, level schedule) that the
// HIP numeric kernels replay for every delta-shift refactorisation.
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

namespace okkt {

// cancel != nullptr: checked at every pivot; when it is set the ordering is abandoned and `order` comes back empty
void amd_order(int n, const std::vector<int64_t>& ap, const std::vector<int>& ai,
               std::vector<int>& order, const std::atomic<bool>* cancel = nullptr);

// level-structure nested dissection (nd.cpp): the parallel ordering for path-like graphs
void level_nd_order(int n, const std::vector<int64_t>& gp, const std::vector<int>& gi, int leaf, std::vector<int>& order);

// multilevel nested dissection (mlnd.cpp): heavy-edge coarsening, FM-refined bisections, minimum-vertex-cover separators,
// minimum degree on the pieces of at most `leaf` vertices; the top two levels keep the best of `ntrial_top` bisections.
// Leaves `order` empty when the graph is beyond its 32-bit offsets (the caller keeps minimum degree).
// top_sep != nullptr: receives the size of the top-level separator (-1 when the graph was not bisected)
void ml_nd_order(int n, const std::vector<int64_t>& gp, const std::vector<int>& gi, int leaf, int ntrial_top, std::vector<int>& order, int* top_sep = nullptr);

struct SymbolicOptions {
  int ordering = 0;        // 0 = AMD, switching to level-structure nested dissection when the AMD tree is a path of small
                           // fronts (see analyze_pattern); 1 = natural, 2 = user permutation, 3 = AMD always, 4 = level-structure nested dissection always, 5 = multilevel nested dissection always
  int nd_leaf = 24;        // level-structure nested dissection stops at pieces of this many nodes (round 2: 48)
  int mlnd_leaf = 1500;    // multilevel nested dissection (ordering 5, or chosen by ordering 0): pieces ordered by minimum degree
  int mlnd_trials = 3;     // bisections tried (different seeds, side by side on host threads) on the top two levels
  int relax_always = 64;   // merge a child into its parent when the merged width <= this (one LDS-resident front instead of a chain of launches / loop trips)
  int relax_small = 256;   // ... or when width <= relax_small and zero fraction < relax_small_frac.  Round 3 sweep (scripts/relax_sweep.sh,
                           // 40 settings x 3 configurations; round 2 had 128 / 0.5): a level of fronts with 128 pivot columns costs 0.5 ms of
                           // assembly and launch chain whatever its flops, and a stricter zero fraction with a wider limit removes levels
                           // without adding flops -- S-metric 15 -> 11 levels (23.0 + 1.86 -> 21.8 + 1.60 ms), S-C3 10 -> 8 (3.95 + 0.93 ->
                           // 3.69 + 0.86).  512 / 0.3 is faster still on S-C5 (-12 %) but its wide fronts take the solves through the
                           // 2048-column inverses: forward error 3.5e-8 instead of 2e-9 (scripts/forward_error.py), so it is not the default
  double relax_small_frac = 0.25;
  int relax_mid = 96;
  double relax_mid_frac = 0.15;
  double relax_any_frac = 0.03;
  int small_front_max = 64;  // fronts of order <= this take the LDS-resident kernel
  int panel_nb = 128;        // block-column width of the big-front kernels
};

// One supernode == one frontal matrix of order f = k + r:
//   k pivot columns  [col0, col0+k)   (permuted numbering, contiguous)
//   r off-diagonal rows (ancestors' columns), sorted ascending
// The front is a dense column-major f x f buffer at front_pos (lower triangle used);
// its first k columns become the L panel (unit diagonal implied, D on the diagonal),
// its trailing r x r block is the contribution (update) block passed to the parent.
struct Symbolic {
  int64_t n = 0;          // matrix order
  int64_t nnz_in = 0;     // entries in the caller's CSC (all, incl. ignored upper ones)
  int64_t nnz_lower = 0;  // entries with row >= col (the ones that are used)
  bool has_duplicates = false;

  int64_t top_separator = -1;   // multilevel dissection: vertices of the top-level separator (-1: no bisection)
  bool amd_skipped = false;     // automatic ordering: minimum degree was abandoned because the dissection's top separator is small (no flop comparison)
  double flops_other = 0;       // automatic ordering: factor flops of the candidate that lost the comparison (0: none / skipped)

  std::vector<int> perm;   // perm[new] = old
  std::vector<int> iperm;  // iperm[old] = new
  std::vector<int> parent; // column elimination tree (permuted numbering)
  std::vector<int> colcount;  // nnz of each column of L incl. diagonal (before relaxation)

  int nsuper = 0;
  std::vector<int> sn_col0;     // [nsuper+1] first column of each supernode
  std::vector<int> sn_parent;   // [nsuper] parent supernode or -1
  std::vector<int> sn_level;    // [nsuper] height above the leaves
  std::vector<int> col2sn;      // [n]
  std::vector<int64_t> row_ptr; // [nsuper+1] into rows
  std::vector<int> rows;        // front row lists (global permuted indices), first k are the columns
  std::vector<int64_t> front_pos;  // [nsuper+1] offsets (in doubles) of the f x f buffers
  std::vector<int64_t> child_ptr;  // [nsuper+1]
  std::vector<int> children;       // child supernodes grouped by parent, ascending
  std::vector<int64_t> rel_ptr;    // [nsuper+1] into rel (length r of each supernode)
  std::vector<int> rel;            // position of each off-diagonal row inside the parent's front
  std::vector<int64_t> cv_pos;     // [nsuper+1] offsets of the solve contribution vectors (length r)

  // scatter of the caller's values into the fronts
  std::vector<int64_t> amap;     // [nnz_in] destination offset in the front arena, -1 = ignored
  std::vector<int64_t> diag_pos; // [n] arena offset of the diagonal entry of ORIGINAL index i
  // the same, grouped by destination supernode (for the fused assemble-in-LDS kernel)
  std::vector<int64_t> aent_ptr; // [nsuper+1]
  std::vector<int64_t> aent_src; // source index into nzval
  std::vector<int> aent_dst;     // local offset lrow + lcol * f inside the front

  // level schedule: supernodes ordered by level
  int nlevels = 0;
  std::vector<int> level_ptr;    // [nlevels+1] into level_sn
  std::vector<int> level_sn;

  // statistics (algorithmic work)
  int64_t nnzL = 0;          // sum_j colcount_j (exact structure, no relaxation zeros)
  int64_t nnzL_stored = 0;   // sum_s (f*k - k(k-1)/2): panel entries incl. relaxation zeros
  double flops_exact = 0;    // sum_j colcount_j^2
  double flops_stored = 0;   // dense-front flops actually executed
  int64_t arena_doubles = 0; // sum_s f^2
  int64_t sum_r = 0;
  int max_front = 0;
  int ordering_used = 0;     // 0 = AMD, 1 = natural, 2 = user, 4 = level-structure nested dissection
  int64_t critical_pivots = 0;   // pivots on the longest leaf-to-root path of the supernodal tree (n for a path)
  uint64_t pattern_hash = 0;

  // subtree-to-GPU partition (multi-GPU row of SURVEY 8e): owner part of every supernode, -1 = "top"
  // (ancestors of the cut, factored by part 0 after the contribution blocks of the cut have arrived)
  int nparts = 1;
  std::vector<int> sn_owner;          // [nsuper]
  std::vector<int> boundary;          // subtree roots whose parent is a top node, ascending
  std::vector<int64_t> boundary_cb;   // [nboundary+1] offsets (doubles) of their r x r blocks in the exchange buffer
  std::vector<int64_t> boundary_cv;   // [nboundary+1] offsets of their length-r vectors
  std::vector<double> part_flops;     // [nparts] dense-front flops of the subtrees of each part
  double top_flops = 0;
};

// assigns disjoint elimination-tree subtrees to nparts parts (greedy: split the heaviest subtree until the
// pieces are small enough to balance, then longest-processing-time bin packing); deterministic
void partition_tree(Symbolic& S, int nparts);


// colptr/rowval: CSC of a square matrix in either index base; only row >= col is used.
// user_perm (size n, perm[new]=old, 0-based) is read when opts.ordering == 2.
// Returns "" on success, otherwise an error message.
std::string analyze_pattern(int64_t n, const int64_t* colptr, const int64_t* rowval,
                            int index_base, const SymbolicOptions& opts,
                            const int64_t* user_perm, Symbolic& S);

uint64_t hash_pattern(int64_t n, const int64_t* colptr, const int64_t* rowval);

}  // namespace okkt
