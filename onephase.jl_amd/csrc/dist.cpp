// C ABI, multi-GPU level (include/okkt.h): subtree-to-GPU sharding of one factorisation.
//
// Every rank (one process per GPU) analyses the same pattern, calls okkt_dist_set_partition with its own
// part id and then walks the phases below; the caller moves three flat buffers between ranks with its
// collective of choice (RCCL reduce / broadcast through torch.distributed in bench.py and
// onephase.jl_amd/distributed.py).  There is no reference counterpart: the reference is single-process
// (SURVEY.md section 8e); results are checked against the single-GPU path.
//
//   factor:  local subtrees -> pack contribution blocks of the cut -> [reduce to part 0] -> unpack ->
//            top of the tree on part 0 -> pivot counts [all-reduce] -> okkt_dist_finish
//   solve :  forward on local subtrees -> pack contribution vectors -> [reduce to part 0] -> unpack ->
//            forward+backward on the top -> x of the top columns [broadcast] -> backward on local subtrees ->
//            owned part of the solution, original order [reduce / all-reduce]
#include <cstring>

#include "solver.h"

using namespace okkt;

namespace {
int need_dist(okkt_solver_s* h) {
  int rc = solver_ensure_numeric(h);
  if (rc != OKKT_OK) return rc;
  return OKKT_OK;
}
int sync_or_fail(okkt_solver_s* h, const char* what) {
  hipError_t he = hipStreamSynchronize(h->stream);
  if (he != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(he));
  return OKKT_OK;
}
}  // namespace

extern "C" {

int okkt_dist_set_partition(okkt_handle h, int nparts, int part_id) {
  if (!h || nparts < 1 || part_id < 0 || part_id >= nparts) return OKKT_ERR_INVALID;
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "okkt_analyze has not been called");
  if (h->numeric_ready) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    numeric_release(h->N);
    h->numeric_ready = false;
  }
  h->factored = false;
  partition_tree(h->S, nparts);
  h->part_id = part_id;
  return OKKT_OK;
}

int okkt_dist_info(okkt_handle h, int64_t* cb_doubles, int64_t* cv_doubles, int64_t* n_boundary,
                   double* part_flops_out, double* top_flops_out) {
  if (!h) return OKKT_ERR_INVALID;
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "not analysed");
  const Symbolic& S = h->S;
  if (cb_doubles) *cb_doubles = S.boundary_cb.empty() ? 0 : S.boundary_cb.back();
  if (cv_doubles) *cv_doubles = S.boundary_cv.empty() ? 0 : S.boundary_cv.back();
  if (n_boundary) *n_boundary = (int64_t)S.boundary.size();
  if (part_flops_out) for (int p = 0; p < S.nparts && p < (int)S.part_flops.size(); ++p) part_flops_out[p] = S.part_flops[p];
  if (top_flops_out) *top_flops_out = S.top_flops;
  return OKKT_OK;
}

int okkt_dist_get_owner(okkt_handle h, int64_t* sn_owner_out, int64_t* col_owner_out, int64_t* sn_parent_out) {
  if (!h) return OKKT_ERR_INVALID;
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "not analysed");
  const Symbolic& S = h->S;
  for (int s = 0; s < S.nsuper; ++s) {
    const int o = (int)S.sn_owner.size() == S.nsuper ? S.sn_owner[s] : 0;
    if (sn_owner_out) sn_owner_out[s] = o;
    if (sn_parent_out) sn_parent_out[s] = S.sn_parent[s];
    if (col_owner_out) for (int j = S.sn_col0[s]; j < S.sn_col0[s + 1]; ++j) col_owner_out[j] = o;
  }
  return OKKT_OK;
}

int okkt_dist_factor_local(okkt_handle h, const double* d_nzval, int64_t n, int64_t m, int sym_kind) {
  if (!h || (!d_nzval && h->S.nnz_in > 0)) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  if (n < 0 || m < 0 || n + m != h->S.n) return solver_set_error(h, OKKT_ERR_INVALID, "n + m does not match the analysed dimension");
  if (sym_kind != OKKT_SYM_DEFINITE && sym_kind != OKKT_SYM_SYMMETRIC) return solver_set_error(h, OKKT_ERR_INVALID, "unknown sym_kind");
  h->dist_vals = d_nzval;
  h->dist_n = n; h->dist_m = m; h->dist_kind = sym_kind;
  h->dist_tol = sym_kind == OKKT_SYM_DEFINITE ? 0.0 : h->opts.inertia_tol;
  h->factored = false;
  // a handle that was used with early exit before keeps no stale state: the sharded path always runs to the end
  h->N.early_check = false;
  h->N.early_device = false;
  h->N.early_exited = false;
  std::string e = numeric_factor_enqueue(h->N, d_nzval, h->dist_tol, 0, true);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "local factorisation");
}

int okkt_dist_cb(okkt_handle h, double* d_buf, int unpack) {
  if (!h || !d_buf) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  std::string e = numeric_dist_pack(h->N, 0, unpack, d_buf);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "contribution-block exchange");
}

int okkt_dist_factor_top(okkt_handle h) {
  if (!h) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  std::string e = numeric_factor_enqueue(h->N, h->dist_vals, h->dist_tol, 1, false);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "top factorisation");
}

int okkt_dist_counts(okkt_handle h, int64_t out[4]) {
  if (!h || !out) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  unsigned long long cnt[5];
  if (!numeric_read_counts(h->N, h->stream, cnt).empty())
    return solver_set_error(h, OKKT_ERR_HIP, "download of the pivot counts failed");
  for (int i = 0; i < 4; ++i) out[i] = (int64_t)cnt[i];
  return OKKT_OK;
}

int okkt_dist_finish(okkt_handle h, const int64_t total[4]) {
  if (!h || !total) return OKKT_ERR_INVALID;
  h->factored = true;
  if (total[0] + total[1] + total[2] + total[3] != h->S.n)
    return solver_set_error(h, OKKT_ERR_INTERNAL, "pivot counts do not add up to the matrix order");
  if (total[3] > 0) return 0;
  if (h->dist_kind == OKKT_SYM_DEFINITE) return total[0] == h->dist_n ? 1 : 0;
  return (total[0] == h->dist_n && total[1] == h->dist_m) ? 1 : 0;
}

int okkt_dist_solve_begin(okkt_handle h, const double* d_rhs) {
  if (!h || !d_rhs) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  if (!h->factored) return solver_set_error(h, OKKT_ERR_INVALID, "solve called before a factorisation");
  solve_permute_in(h->N, d_rhs, h->S.n, 1, 1);
  std::string e = solve_fwd_enqueue(h->N, 0, 1);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "local forward solve");
}

int okkt_dist_cv(okkt_handle h, double* d_buf, int unpack) {
  if (!h || !d_buf) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  std::string e = numeric_dist_pack(h->N, 1, unpack, d_buf);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "contribution-vector exchange");
}

int okkt_dist_solve_top(okkt_handle h) {
  if (!h) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  std::string e = solve_fwd_enqueue(h->N, 1, 1);
  if (e.empty()) e = solve_bwd_enqueue(h->N, 1, 1);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "top solve");
}

int okkt_dist_x(okkt_handle h, double* d_buf, int mode) {
  if (!h || !d_buf || mode < 0 || mode > 2) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  std::string e = numeric_dist_x(h->N, mode, d_buf);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "solution exchange");
}

int okkt_dist_solve_end(okkt_handle h) {
  if (!h) return OKKT_ERR_INVALID;
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  std::string e = solve_bwd_enqueue(h->N, 0, 1);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  return sync_or_fail(h, "local backward solve");
}

}  // extern "C"
