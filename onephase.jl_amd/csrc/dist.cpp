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
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>

#include <rccl/rccl.h>      // types and enums only: the entry points are resolved with dlsym at run time

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
  unsigned long long cnt[6];
  if (!numeric_read_counts(h->N, h->stream, cnt).empty())
    return solver_set_error(h, OKKT_ERR_HIP, "download of the pivot counts failed");
  for (int i = 0; i < 4; ++i) out[i] = (int64_t)cnt[i];
  // a hand-off that ran into its bound leaves incomplete pivot counts: an error of its own, never an inertia failure the delta loop
  // would answer with a larger shift (advisor, round 5)
  if (cnt[5] != 0) return solver_set_error(h, OKKT_ERR_INTERNAL, "a hand-off inside a launch timed out");
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

// ---- RCCL transport ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
RcclApi& rccl() {
  static RcclApi* api = new RcclApi;      // never destructed: no ordering problem at process exit
  static std::once_flag once;
  std::call_once(once, [] {
    RcclApi& a = *api;
    const char* names[] = {getenv("OKKT_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // A copy that is in the process already (torch brings its own librccl) is taken first: two RCCL libraries in one process ended in
    // "double free or corruption" at exit on the ROCm 7.2 boxes (this library had opened /opt/rocm's copy with RTLD_GLOBAL, a later
    // `import torch` loaded the bundled one on top of its symbols).  RTLD_LOCAL: nothing of RCCL is exported to later loads.
    for (int pass = 0; pass < 2 && !a.lib; ++pass)
      for (const char* nm : names) {
        if (!nm) continue;
        a.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (a.lib) break;
      }
    if (!a.lib) { a.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
#define OKKT_SYM(field, name) a.field = (decltype(a.field))dlsym(a.lib, name); if (!a.field) a.err = std::string("librccl lacks ") + name
    OKKT_SYM(GetUniqueId, "ncclGetUniqueId");
    OKKT_SYM(CommInitRank, "ncclCommInitRank");
    OKKT_SYM(CommDestroy, "ncclCommDestroy");
    OKKT_SYM(Reduce, "ncclReduce");
    OKKT_SYM(Broadcast, "ncclBroadcast");
    OKKT_SYM(AllReduce, "ncclAllReduce");
    OKKT_SYM(GetErrorString, "ncclGetErrorString");
#undef OKKT_SYM
  });
  return *api;
}
int rccl_fail(okkt_solver_s* h, const char* what, ncclResult_t r) {
  RcclApi& a = rccl();
  return solver_set_error(h, OKKT_ERR_HIP, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
}
#define OKKT_NCCL(h, what, expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) return rccl_fail(h, what, r__); } while (0)
int dist_free_buffers(okkt_solver_s* h) {
  for (void* p : {(void*)h->dist_cb, (void*)h->dist_cv, (void*)h->dist_x, (void*)h->dist_counts}) if (p) (void)hipFree(p);
  h->dist_cb = h->dist_cv = h->dist_x = nullptr;
  h->dist_cb_cap = h->dist_cv_cap = h->dist_x_cap = 0;
  h->dist_counts = nullptr;
  return OKKT_OK;
}
}  // namespace

int okkt_dist_unique_id(void* id_out) {
  if (!id_out) return OKKT_ERR_INVALID;
  RcclApi& a = rccl();
  if (!a.err.empty() || !a.GetUniqueId) return OKKT_ERR_NO_DEVICE;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (a.GetUniqueId(&id) != ncclSuccess) return OKKT_ERR_HIP;
  std::memcpy(id_out, &id, sizeof(id));
  return OKKT_OK;
}

int okkt_dist_comm_init(okkt_handle h, int nranks, int rank, const void* id) {
  if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return OKKT_ERR_INVALID;
  RcclApi& a = rccl();
  if (!a.err.empty()) return solver_set_error(h, OKKT_ERR_NO_DEVICE, a.err);
  if (!h->device_ready) return solver_set_error(h, OKKT_ERR_NO_DEVICE, "no HIP device");
  if (!h->analyzed || h->S.nparts != nranks || h->part_id != rank)
    return solver_set_error(h, OKKT_ERR_INVALID, "okkt_dist_set_partition(nranks, rank) must precede okkt_dist_comm_init");
  if (hipSetDevice(h->device) != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, "hipSetDevice failed");
  if (h->rccl_comm) { (void)a.CommDestroy((ncclComm_t)h->rccl_comm); h->rccl_comm = nullptr; }
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  OKKT_NCCL(h, "ncclCommInitRank", a.CommInitRank(&comm, nranks, uid, rank));
  h->rccl_comm = comm; h->rccl_nranks = nranks; h->rccl_rank = rank;
  // exchange buffers (contribution blocks / vectors of the cut, solution pieces): sized by the partition
  dist_free_buffers(h);
  const size_t cb = h->S.boundary_cb.empty() ? 0 : (size_t)h->S.boundary_cb.back(), cv = h->S.boundary_cv.empty() ? 0 : (size_t)h->S.boundary_cv.back();
  if (hipMalloc((void**)&h->dist_cb, std::max<size_t>(cb, 1) * 8) != hipSuccess || hipMalloc((void**)&h->dist_cv, std::max<size_t>(cv, 1) * 8) != hipSuccess ||
      hipMalloc((void**)&h->dist_x, std::max<size_t>((size_t)h->S.n, 1) * 8) != hipSuccess || hipMalloc((void**)&h->dist_counts, 8 * sizeof(long long)) != hipSuccess)
    return solver_set_error(h, OKKT_ERR_ALLOC, "exchange buffers");
  h->dist_cb_cap = cb; h->dist_cv_cap = cv; h->dist_x_cap = (size_t)h->S.n;
  // Every rank analysed the pattern on its own: the plans must be the same plan.  A digest of the permutation, the supernode
  // partition and the cut is compared across the ranks (max == min of the digest's two halves); ranks that disagree would pack
  // contribution blocks of different sizes and hang in the first collective (advisor, round 5).
  if (nranks > 1) {
    uint64_t dg = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { dg ^= v; dg *= 1099511628211ull; dg = (dg << 23) | (dg >> 41); };
    for (int v : h->S.perm) mix((uint64_t)(uint32_t)v);
    for (int v : h->S.sn_col0) mix((uint64_t)(uint32_t)v);
    for (int v : h->S.sn_owner) mix((uint64_t)(uint32_t)v);
    for (int v : h->S.boundary) mix((uint64_t)(uint32_t)v);
    long long hv[4] = {(long long)(dg >> 32), (long long)(dg & 0xffffffffull), -(long long)(dg >> 32), -(long long)(dg & 0xffffffffull)};   // max of (x, -x) = (max, -min)
    if (hipMemcpyAsync(h->dist_counts, hv, sizeof(hv), hipMemcpyHostToDevice, h->stream) != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, "digest upload");
    OKKT_NCCL(h, "ncclAllReduce(plan digest)", a.AllReduce(h->dist_counts, h->dist_counts, 4, ncclInt64, ncclMax, comm, h->stream));
    long long rv[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(rv, h->dist_counts, sizeof(rv), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
      return solver_set_error(h, OKKT_ERR_HIP, "digest download");
    if (rv[0] != -rv[2] || rv[1] != -rv[3])
      return solver_set_error(h, OKKT_ERR_INVALID, "the ranks analysed different plans (permutation / partition digests differ): use the same options and pattern on every rank");
  }
  return OKKT_OK;
}

// the exchange buffers were sized for the partition that was current at okkt_dist_comm_init: a re-analysis or a new
// okkt_dist_set_partition without a new comm_init must not pack into the old allocation
static int dist_buffers_fit(okkt_handle h) {
  const size_t cb = h->S.boundary_cb.empty() ? 0 : (size_t)h->S.boundary_cb.back(), cv = h->S.boundary_cv.empty() ? 0 : (size_t)h->S.boundary_cv.back();
  if (!h->dist_cb || !h->dist_cv || !h->dist_x || cb > h->dist_cb_cap || cv > h->dist_cv_cap || (size_t)h->S.n > h->dist_x_cap ||
      h->S.nparts != h->rccl_nranks || h->part_id != h->rccl_rank)
    return solver_set_error(h, OKKT_ERR_INVALID, "the partition or the pattern changed after okkt_dist_comm_init: call okkt_dist_comm_init again");
  return OKKT_OK;
}

int okkt_dist_comm_destroy(okkt_handle h) {
  if (!h) return OKKT_ERR_INVALID;
  if (h->rccl_comm) {
    (void)hipStreamSynchronize(h->stream);
    (void)rccl().CommDestroy((ncclComm_t)h->rccl_comm);
    h->rccl_comm = nullptr;
  }
  return dist_free_buffers(h);
}

int okkt_dist_factor(okkt_handle h, const double* d_nzval, int64_t n, int64_t m, int sym_kind, okkt_inertia* inertia_out) {
  if (!h || (!d_nzval && h->S.nnz_in > 0)) return OKKT_ERR_INVALID;
  if (!h->rccl_comm) return solver_set_error(h, OKKT_ERR_INVALID, "okkt_dist_comm_init has not been called");
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  if ((rc = dist_buffers_fit(h)) != OKKT_OK) return rc;
  if (n < 0 || m < 0 || n + m != h->S.n) return solver_set_error(h, OKKT_ERR_INVALID, "n + m does not match the analysed dimension");
  if (sym_kind != OKKT_SYM_DEFINITE && sym_kind != OKKT_SYM_SYMMETRIC) return solver_set_error(h, OKKT_ERR_INVALID, "unknown sym_kind");
  RcclApi& a = rccl();
  ncclComm_t comm = (ncclComm_t)h->rccl_comm;
  hipStream_t st = h->stream;
  h->dist_vals = d_nzval; h->dist_n = n; h->dist_m = m; h->dist_kind = sym_kind;
  h->dist_tol = sym_kind == OKKT_SYM_DEFINITE ? 0.0 : h->opts.inertia_tol;
  h->factored = false;
  h->N.early_check = false; h->N.early_device = false; h->N.early_exited = false;
  (void)hipEventRecord(h->ev0, st);
  // local subtrees -> pack the contribution blocks of the cut (every slot written: zeros where another part owns it) ->
  // reduce to part 0 -> part 0 unpacks and factors the top -> the four pivot counts summed over the parts; all on `st`
  std::string e = numeric_factor_enqueue(h->N, d_nzval, h->dist_tol, 0, true);
  if (e.empty()) e = numeric_dist_pack(h->N, 0, 0, h->dist_cb);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  const size_t cb = h->S.boundary_cb.empty() ? 0 : (size_t)h->S.boundary_cb.back();
  if (cb && h->rccl_nranks > 1) OKKT_NCCL(h, "ncclReduce(contribution blocks)", a.Reduce(h->dist_cb, h->dist_cb, cb, ncclDouble, ncclSum, 0, comm, st));
  if (h->rccl_rank == 0) {
    e = numeric_dist_pack(h->N, 0, 1, h->dist_cb);
    if (e.empty()) e = numeric_factor_enqueue(h->N, d_nzval, h->dist_tol, 1, false);
    if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  }
  numeric_sum_counts_device(h->N, h->dist_counts);
  if (h->rccl_nranks > 1) OKKT_NCCL(h, "ncclAllReduce(pivot counts)", a.AllReduce(h->dist_counts, h->dist_counts, 5, ncclInt64, ncclSum, comm, st));
  long long tot[5] = {0, 0, 0, 0, 0};      // pos, neg, zero, nonfinite, time-outs of in-launch waits (summed over the ranks)
  if (hipMemcpyAsync(tot, h->dist_counts, sizeof(tot), hipMemcpyDeviceToHost, st) != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, "count download");
  (void)hipEventRecord(h->ev1, st);
  rc = sync_or_fail(h, "sharded factorisation");
  if (rc != OKKT_OK) return rc;
  float ms = 0;
  if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_factor_ms = ms;
  if (tot[4] != 0) return solver_set_error(h, OKKT_ERR_INTERNAL, "a hand-off inside a launch timed out on one of the ranks");
  if (inertia_out) { inertia_out->pos = tot[0]; inertia_out->neg = tot[1]; inertia_out->zero = tot[2]; inertia_out->nonfinite = tot[3]; }
  const int64_t t64[4] = {tot[0], tot[1], tot[2], tot[3]};
  return okkt_dist_finish(h, t64);
}

int okkt_dist_solve(okkt_handle h, const double* d_rhs, double* d_sol) {
  if (!h || !d_rhs || !d_sol) return OKKT_ERR_INVALID;
  if (!h->rccl_comm) return solver_set_error(h, OKKT_ERR_INVALID, "okkt_dist_comm_init has not been called");
  int rc = need_dist(h);
  if (rc != OKKT_OK) return rc;
  if ((rc = dist_buffers_fit(h)) != OKKT_OK) return rc;
  if (!h->factored) return solver_set_error(h, OKKT_ERR_INVALID, "solve called before a factorisation");
  RcclApi& a = rccl();
  ncclComm_t comm = (ncclComm_t)h->rccl_comm;
  hipStream_t st = h->stream;
  const bool multi = h->rccl_nranks > 1;
  (void)hipEventRecord(h->ev0, st);
  // forward on the subtrees -> contribution vectors of the cut reduced to part 0 -> forward + backward on the top ->
  // separator solution broadcast -> backward on the subtrees -> owned solution pieces summed (all-reduce): every rank ends
  // with the whole solution.  One stream, no host synchronisation before the end.
  solve_permute_in(h->N, d_rhs, h->S.n, 1, 1);
  std::string e = solve_fwd_enqueue(h->N, 0, 1);
  if (e.empty()) e = numeric_dist_pack(h->N, 1, 0, h->dist_cv);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  const size_t cv = h->S.boundary_cv.empty() ? 0 : (size_t)h->S.boundary_cv.back();
  if (cv && multi) OKKT_NCCL(h, "ncclReduce(contribution vectors)", a.Reduce(h->dist_cv, h->dist_cv, cv, ncclDouble, ncclSum, 0, comm, st));
  if (h->rccl_rank == 0) {
    e = numeric_dist_pack(h->N, 1, 1, h->dist_cv);
    if (e.empty()) e = solve_fwd_enqueue(h->N, 1, 1);
    if (e.empty()) e = solve_bwd_enqueue(h->N, 1, 1);
    if (e.empty()) e = numeric_dist_x(h->N, 0, h->dist_x);
    if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  }
  // only the top's columns travel (round 4 sent all n doubles for a separator of a few hundred)
  if (multi && h->N.n_top_cols > 0) OKKT_NCCL(h, "ncclBroadcast(separator solution)", a.Broadcast(h->dist_x, h->dist_x, (size_t)h->N.n_top_cols, ncclDouble, 0, comm, st));
  if (multi || h->rccl_rank != 0) e = numeric_dist_x(h->N, 1, h->dist_x);
  if (e.empty()) e = solve_bwd_enqueue(h->N, 0, 1);
  if (e.empty()) e = numeric_dist_x(h->N, 2, d_sol);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  if (multi) OKKT_NCCL(h, "ncclAllReduce(solution)", a.AllReduce(d_sol, d_sol, (size_t)h->S.n, ncclDouble, ncclSum, comm, st));
  (void)hipEventRecord(h->ev1, st);
  rc = sync_or_fail(h, "sharded solve");
  if (rc != OKKT_OK) return rc;
  float ms = 0;
  if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_solve_ms = ms;
  return OKKT_OK;
}

}  // extern "C"
