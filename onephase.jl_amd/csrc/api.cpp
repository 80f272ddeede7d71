// C ABI, linear-solver level (include/okkt.h): the entry points a
// `linear_solver_HIP <: abstract_linear_system_solver` binds in place of linear_solver_JULIA
// (/root/reference/src/linear_system_solvers/julia.jl).  No exception leaves this file.
#include <mutex>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>

#include "solver.h"

using namespace okkt;

namespace okkt {

int solver_set_error(okkt_solver_s* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

static int ensure_device(okkt_solver_s* h) {
  if (h->opts.host_symbolic_only) return solver_set_error(h, OKKT_ERR_NO_DEVICE, "handle was created with host_symbolic_only");
  if (h->device_ready) {
    if (hipSetDevice(h->device) != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, "hipSetDevice failed");
    return OKKT_OK;
  }
  return solver_set_error(h, OKKT_ERR_NO_DEVICE, "no HIP device");
}

int solver_ensure_numeric(okkt_solver_s* h) {
  int rc = ensure_device(h);
  if (rc != OKKT_OK) return rc;
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "okkt_analyze has not been called");
  if (h->numeric_ready) return OKKT_OK;
  h->N.part_id = h->part_id;
  h->N.stream_masked = h->stream_masked;     // before the set-up: the lanes of the plan are given streams there
  h->N.stream_panel = h->stream_panel;
  h->N.stream_aux = h->stream_aux;
  if (const char* fl = getenv("OKKT_FLOW")) h->N.flow = atoi(fl);   // read by the set-up
  if (const char* df = getenv("OKKT_DATAFLOW")) h->N.dataflow = atoi(df);
  if (const char* sf = getenv("OKKT_SOLVE_FLOW")) h->N.solve_flow = atoi(sf);   // read by the set-up (it sizes the partial-product buffers)
  if (const char* sw = getenv("OKKT_SOLVE_FUSE_WIDE_MAX")) h->N.solve_fuse_wide_max = atoi(sw);   // read by the set-up (the experiment keeps the explicit inverses for every front)
  std::string e = numeric_setup(h->S, h->sopts, h->stream, h->N);
  if (const char* sh = getenv("OKKT_SPLIT_HEAD")) h->N.split_head = atoi(sh);
  if (const char* d2 = getenv("OKKT_DIAG2")) h->N.diag2 = atoi(d2);
  if (const char* fd = getenv("OKKT_FUSE_DIAG_TRSM")) h->N.fuse_diag_trsm = atoi(fd);
  if (const char* ss = getenv("OKKT_SOLVE_SPLIT_SMALL")) h->N.solve_split_small = atoi(ss);
  if (const char* su = getenv("OKKT_SOLVE_FUSE")) h->N.solve_fuse = atoi(su);
  if (const char* mt = getenv("OKKT_LA_MIN_TILES")) h->N.la_min_tiles = atoi(mt);
  if (!e.empty()) { numeric_release(h->N); return solver_set_error(h, OKKT_ERR_HIP, e); }
  h->numeric_ready = true;
  return OKKT_OK;
}

int solver_factor_device(okkt_solver_s* h, const double* d_vals, int64_t n, int64_t m, int sym_kind,
                         okkt_inertia* out) {
  int rc = solver_ensure_numeric(h);
  if (rc != OKKT_OK) return rc;
  if (n < 0 || m < 0 || n + m != h->S.n) return solver_set_error(h, OKKT_ERR_INVALID, "n + m does not match the analysed dimension");
  if (sym_kind != OKKT_SYM_DEFINITE && sym_kind != OKKT_SYM_SYMMETRIC) return solver_set_error(h, OKKT_ERR_INVALID, "unknown sym_kind");
  if (sym_kind == OKKT_SYM_DEFINITE && m != 0) return solver_set_error(h, OKKT_ERR_INVALID, ":definite requires m == 0 (julia.jl:30)");
  const double tol = sym_kind == OKKT_SYM_DEFINITE ? 0.0 : h->opts.inertia_tol;
  h->factored = false;
  h->N.early_check = h->early_exit;
  h->N.early_device = h->early_exit && h->last_failed;   // the previous factorisation failed the inertia: this one is a retry
  h->N.early_n = n;
  h->N.early_m = m;
  (void)hipEventRecord(h->ev0, h->stream);
  std::string e = numeric_factor_enqueue(h->N, d_vals, tol);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
  (void)hipEventRecord(h->ev1, h->stream);
  unsigned long long cnt[6];
  e = numeric_read_counts(h->N, h->stream, cnt);
  if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, std::string("numeric factorisation failed: ") + e);
  if (cnt[5] != 0)      // distinct from an inertia failure: callers of the delta loop must not shift and retry on it
    return solver_set_error(h, OKKT_ERR_INTERNAL, "a hand-off inside a launch timed out: the pivot counts are incomplete and there is no factor");
  float ms = 0;
  if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_factor_ms = ms;
  okkt_inertia in;
  in.pos = (int64_t)cnt[0]; in.neg = (int64_t)cnt[1]; in.zero = (int64_t)cnt[2]; in.nonfinite = (int64_t)cnt[3];
  if (out) *out = in;
  h->last_failed = true;
  if (h->N.early_exited || cnt[4] != 0) return 0;   // wrong inertia decided before the end: counts are partial, no factor to solve with
  h->factored = true;
  if (in.pos + in.neg + in.zero + in.nonfinite != h->S.n)
    return solver_set_error(h, OKKT_ERR_INTERNAL, "pivot counts do not add up to the matrix order");
  if (in.nonfinite > 0) return 0;                       // julia.jl:77-89
  const int flag = sym_kind == OKKT_SYM_DEFINITE ? (in.pos == n ? 1 : 0)   // PosDefException <=> some pivot <= 0
                                                 : ((in.pos == n && in.neg == m) ? 1 : 0);   // linear_system_solvers.jl:73-74
  h->last_failed = flag == 0;
  return flag;
}

int solver_solve_enqueue(okkt_solver_s* h, const double* d_rhs, double* d_sol, int64_t nrhs, bool accumulate) {
  int rc = solver_ensure_numeric(h);
  if (rc != OKKT_OK) return rc;
  if (!h->factored) return solver_set_error(h, OKKT_ERR_INVALID, "solve called before a factorisation");
  if (nrhs < 0) return solver_set_error(h, OKKT_ERR_INVALID, "nrhs < 0");
  // batches of up to kMaxRhs right-hand sides: one pass over L per batch (R = 1, 2 or 4 kernels; three are padded to four)
  for (int64_t r = 0; r < nrhs;) {
    const int nr = (int)std::min<int64_t>(nrhs - r, kMaxRhs);
    const int R = nr >= 3 ? 4 : nr;
    solve_permute_in(h->N, d_rhs + r * h->S.n, h->S.n, nr, R);
    std::string e = numeric_solve_enqueue(h->N, R);
    if (!e.empty()) return solver_set_error(h, OKKT_ERR_HIP, e);
    solve_permute_out(h->N, d_sol + r * h->S.n, h->S.n, nr, R, accumulate);
    r += nr;
  }
  return OKKT_OK;
}

int solver_solve_device(okkt_solver_s* h, const double* d_rhs, double* d_sol, int64_t nrhs) {
  (void)hipEventRecord(h->ev0, h->stream);
  int rc = solver_solve_enqueue(h, d_rhs, d_sol, nrhs, false);
  if (rc != OKKT_OK) return rc;
  (void)hipEventRecord(h->ev1, h->stream);
  hipError_t he = hipStreamSynchronize(h->stream);
  if (he != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, std::string("solve failed: ") + hipGetErrorString(he));
  float ms = 0;
  if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_solve_ms = ms;
  return OKKT_OK;
}

}  // namespace okkt

extern "C" {

const char* okkt_version(void) {
  // (the product library carries none of the experimental roles of the dataflow kernel; libonephase_kkt_exp.so carries all three)
  static const std::string v = std::string("onephase-kkt-mi355x 0.1 (gfx950; dataflow roles: ") + okkt::df_build_flags() + ")";
  return v.c_str();
}

int okkt_default_opts(okkt_opts* o) {
  if (!o) return OKKT_ERR_INVALID;
  std::memset(o, 0, sizeof(*o));
  SymbolicOptions d;
  o->device = -1;
  o->host_symbolic_only = 0;
  o->ordering = 0;
  o->relax_always = d.relax_always;
  o->relax_small = d.relax_small;
  o->relax_mid = d.relax_mid;
  o->relax_small_frac = d.relax_small_frac;
  o->relax_mid_frac = d.relax_mid_frac;
  o->relax_any_frac = d.relax_any_frac;
  o->inertia_tol = 1e-20;
  o->small_front_max = 128;
  o->panel_nb = 128;
  o->early_exit = 0;
  return OKKT_OK;
}

// Stream sets are pooled per (device, look-ahead, reserved CUs) for the life of the process: a handle takes a set and
// gives it back in okkt_destroy.  Measured on MI355X: streams created after another set was destroyed (a second handle
// in the same process) ran the look-ahead schedule 13 % slower (62.9 vs 54.6 ms on the Schur-shape S-metric system)
// -- the runtime's hardware-queue assignment of later streams differs -- while a reused set keeps the first timing.
struct StreamSet {
  int device = -1, la = 0, reserved = 0, seq = 0;   // seq: order of creation in this process (1, 2, ...)
  hipStream_t stream = nullptr, masked = nullptr, panel = nullptr, aux = nullptr;
};
// heap objects that are never destructed: no static-destruction order to get wrong at process exit
static std::mutex& pool_mutex() { static std::mutex* m = new std::mutex; return *m; }
static std::vector<StreamSet>& pool_free() { static std::vector<StreamSet>* v = new std::vector<StreamSet>; return *v; }
static std::vector<StreamSet>& pool_all() { static std::vector<StreamSet>* v = new std::vector<StreamSet>; return *v; }
static bool g_pool_closed = false;

// At exit every stream of the pool is destroyed (in use or not): CU-masked and priority streams that are still alive
// when rocprofv3's tool library tears down crash it (exit code 139 after the traces were written); the handler is
// registered at the first okkt_create, i.e. after HIP's and the profiler's own, so it runs before them.
static void pool_close() {
  std::lock_guard<std::mutex> lock(pool_mutex());
  g_pool_closed = true;
  for (const StreamSet& set : pool_all()) {
    if (hipSetDevice(set.device) != hipSuccess) continue;
    for (hipStream_t q : {set.aux, set.panel, set.masked, set.stream})
      if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
  }
  pool_all().clear();
  pool_free().clear();
}

static bool take_stream_set(int device, int la, int reserved, StreamSet* out) {
  std::lock_guard<std::mutex> lock(pool_mutex());
  std::vector<StreamSet>& fr = pool_free();
  // the OLDEST matching set: later sets share the hardware queues of the earlier ones (a part of the sharded model measured on
  // the third set of a process ran 2.5x slower than on the first), so a lone handle should always get the first one back
  long best = -1;
  for (size_t q = 0; q < fr.size(); ++q)
    if (fr[q].device == device && fr[q].la == la && fr[q].reserved == reserved && (best < 0 || fr[q].seq < fr[(size_t)best].seq)) best = (long)q;
  if (best < 0) return false;
  *out = fr[(size_t)best];
  fr.erase(fr.begin() + best);
  return true;
}
static int pool_size();
static void register_stream_set(const StreamSet& set) {
  std::lock_guard<std::mutex> lock(pool_mutex());
  static bool registered = false;
  if (!registered) { registered = true; (void)atexit(pool_close); }
  pool_all().push_back(set);
  pool_all().back().seq = (int)pool_all().size();
}
static int pool_size() { std::lock_guard<std::mutex> lock(pool_mutex()); return (int)pool_all().size(); }
static void give_stream_set(const StreamSet& set) {
  std::lock_guard<std::mutex> lock(pool_mutex());
  if (!g_pool_closed) pool_free().push_back(set);
}

int okkt_create(okkt_handle* out, const okkt_opts* opts) {
  if (!out) return OKKT_ERR_INVALID;
  *out = nullptr;
  okkt_solver_s* h = new (std::nothrow) okkt_solver_s();
  if (!h) return OKKT_ERR_ALLOC;
  okkt_opts def;
  okkt_default_opts(&def);
  h->opts = opts ? *opts : def;
  okkt_opts& o = h->opts;
  if (o.relax_always <= 0) o.relax_always = def.relax_always;
  if (o.relax_small <= 0) o.relax_small = def.relax_small;
  if (o.relax_mid <= 0) o.relax_mid = def.relax_mid;
  if (o.relax_small_frac <= 0) o.relax_small_frac = def.relax_small_frac;
  if (o.relax_mid_frac <= 0) o.relax_mid_frac = def.relax_mid_frac;
  if (o.relax_any_frac <= 0) o.relax_any_frac = def.relax_any_frac;
  if (o.inertia_tol < 0) o.inertia_tol = def.inertia_tol;
  if (o.small_front_max <= 0) o.small_front_max = def.small_front_max;
  if (o.panel_nb <= 0) o.panel_nb = def.panel_nb;
  h->sopts.ordering = o.ordering;
  h->sopts.relax_always = o.relax_always;
  h->sopts.relax_small = o.relax_small;
  h->sopts.relax_mid = o.relax_mid;
  h->sopts.relax_small_frac = o.relax_small_frac;
  h->sopts.relax_mid_frac = o.relax_mid_frac;
  h->sopts.relax_any_frac = o.relax_any_frac;
  h->sopts.small_front_max = o.small_front_max;
  h->sopts.panel_nb = o.panel_nb;
  h->early_exit = o.early_exit != 0;
  if (!o.host_symbolic_only) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { delete h; return OKKT_ERR_NO_DEVICE; }
    int dev = o.device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= count || hipSetDevice(dev) != hipSuccess) { delete h; return OKKT_ERR_NO_DEVICE; }
    h->device = dev;
    // Look-ahead needs a CU that the trailing update never occupies (k_big_diag wants a whole CU's LDS): levels that
    // use it run on a twin of the handle's stream whose CU mask leaves out the first `reserved` CUs (mask bit b =
    // CU b / 8 of XCD b % 8 on gfx950, probed with scripts/cumask_probe.hip); the panel streams are unmasked and high
    // priority.  Everything else (small fronts, levels without look-ahead, the solves) keeps all CUs.
    const char* ela = getenv("OKKT_LOOKAHEAD");
    const char* ercu = getenv("OKKT_RESERVED_CUS");
    const int la = ela ? atoi(ela) : 1;
    // 32 = one CU of every shader engine of every XCD (mask bit b = CU index b / 8 of XCD b % 8, CU index c in shader engine c % 4:
    // scripts/cumask_map.hip).  The hardware deals workgroups to the shader engines round-robin and IN ORDER: with an uneven mask
    // (round 4: 8 = one CU per XCD) the engine that lost a CU fills up first and the dispatch stalls behind it -- 460 of 496 workgroups
    // resident in scripts/occ_probe.hip, as few as 393 in situ -- so reserving one CU per engine costs the masked stream nothing more
    int reserved = ercu ? atoi(ercu) : 32;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { delete h; return OKKT_ERR_HIP; }
    const int ncu = prop.multiProcessorCount;
    if (reserved < 1) reserved = 1;
    if (reserved > ncu / 2) reserved = ncu / 2;
    StreamSet set;
    if (take_stream_set(dev, la ? 1 : 0, reserved, &set)) {
      h->stream = set.stream; h->stream_masked = set.masked; h->stream_panel = set.panel; h->stream_aux = set.aux;
      h->stream_seq = set.seq;
    } else if (la) {
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      for (int b = reserved; b < ncu; ++b) mask[(size_t)b >> 5] |= 1u << (b & 31);
      int lo = 0, hi = 0;
      if (hipExtStreamCreateWithCUMask(&h->stream_masked, (uint32_t)mask.size(), mask.data()) != hipSuccess ||
          hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
          hipStreamCreateWithPriority(&h->stream_panel, hipStreamNonBlocking, hi) != hipSuccess ||
          hipStreamCreateWithPriority(&h->stream_aux, hipStreamNonBlocking, hi) != hipSuccess) {
        (void)hipGetLastError();
        if (h->stream_masked) { (void)hipStreamDestroy(h->stream_masked); h->stream_masked = nullptr; }
        if (h->stream_panel) { (void)hipStreamDestroy(h->stream_panel); }
        h->stream_panel = nullptr;
        h->stream_aux = nullptr;
      }
    }
    if (!h->stream && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
      // nothing of this set is registered yet: destroy the look-ahead streams that were created above
      for (hipStream_t q : {h->stream_masked, h->stream_panel, h->stream_aux}) if (q) (void)hipStreamDestroy(q);
      delete h;
      return OKKT_ERR_HIP;
    }
    h->stream_la = la ? 1 : 0;
    h->stream_reserved = reserved;
    if (set.device < 0) {   // a new set: the pool owns its streams from now on
      set.device = dev; set.la = h->stream_la; set.reserved = reserved;
      set.stream = h->stream; set.masked = h->stream_masked; set.panel = h->stream_panel; set.aux = h->stream_aux;
      register_stream_set(set);
      h->stream_seq = pool_size();
    }
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
      if (h->ev0) (void)hipEventDestroy(h->ev0);
      give_stream_set(set);      // registered (new or reused): back to the free list for the next handle
      delete h;
      return OKKT_ERR_HIP;
    }
    h->device_ready = true;
  }
  *out = h;
  return OKKT_OK;
}

int okkt_destroy(okkt_handle h) {
  if (!h) return OKKT_ERR_INVALID;
  if (h->rccl_comm || h->dist_cb || h->dist_x) (void)okkt_dist_comm_destroy(h);
  bool closed;
  { std::lock_guard<std::mutex> lock(pool_mutex()); closed = g_pool_closed; }
  if (h->device_ready && closed) {
    // a finalizer that runs after the process-exit handler: every pooled stream has been synchronised and destroyed
    // there, so nothing is in flight; only memory and events are released, no stream is touched
    numeric_release(h->N);
    if (h->d_rhs_stage) (void)hipFree(h->d_rhs_stage);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
  } else if (h->device_ready) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    numeric_release(h->N);
    if (h->d_rhs_stage) (void)hipFree(h->d_rhs_stage);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    // the streams go back to the pool (idle: everything on them was synchronised above or joined into h->stream)
    for (hipStream_t q : {h->stream_masked, h->stream_panel, h->stream_aux})
      if (q) (void)hipStreamSynchronize(q);
    if (h->stream) {
      StreamSet set;
      set.device = h->device; set.la = h->stream_la; set.reserved = h->stream_reserved;
      set.stream = h->stream; set.masked = h->stream_masked; set.panel = h->stream_panel; set.aux = h->stream_aux;
      set.seq = h->stream_seq;
      give_stream_set(set);
    }
  }
  delete h;
  return OKKT_OK;
}

int okkt_set_early_exit(okkt_handle h, int enable) {
  if (!h) return OKKT_ERR_INVALID;
  h->early_exit = enable != 0;
  return OKKT_OK;
}

const char* okkt_last_error(okkt_handle h) { return h ? h->err.c_str() : "null handle"; }

int okkt_set_perm(okkt_handle h, const int64_t* perm, int64_t n) {
  if (!h || !perm || n < 0) return OKKT_ERR_INVALID;
  h->user_perm.assign(perm, perm + n);
  h->analyzed = false;  // force re-analysis with the new permutation
  return OKKT_OK;
}

int okkt_analyze(okkt_handle h, int64_t dim, const int64_t* colptr, const int64_t* rowval, int index_base) {
  if (!h || !colptr || (dim > 0 && !rowval && colptr[dim] != colptr[0])) return OKKT_ERR_INVALID;
  if (dim < 0) return solver_set_error(h, OKKT_ERR_INVALID, "dim < 0");
  try {
    if (h->analyzed && h->S.n == dim) {
      // same pattern as last time?  (the reference rebuilds Q every outer iteration with an
      // identical structure; ls_factor! may therefore call this unconditionally)
      // exact comparison with the analysed pattern (kept on the host): 26 MB of memcmp at S-metric = 2-3 ms per
      // ls_factor!, where the byte-wise hash of the same arrays took 21 ms
      const int64_t nnz = colptr[dim] - colptr[0];
      if (nnz == h->S.nnz_in && (int64_t)h->pat_colptr.size() == dim + 1 && (int64_t)h->pat_rowval.size() == nnz &&
          std::memcmp(h->pat_colptr.data(), colptr, (size_t)(dim + 1) * sizeof(int64_t)) == 0 &&
          (nnz == 0 || std::memcmp(h->pat_rowval.data(), rowval, (size_t)nnz * sizeof(int64_t)) == 0))
        return OKKT_OK;
    }
    if (h->opts.ordering == 2 && (int64_t)h->user_perm.size() != dim)
      return solver_set_error(h, OKKT_ERR_INVALID, "ordering=user: okkt_set_perm must supply dim entries first");
    auto t0 = std::chrono::steady_clock::now();
    if (h->numeric_ready) {
      (void)hipSetDevice(h->device);
      (void)hipStreamSynchronize(h->stream);
      numeric_release(h->N);
      h->numeric_ready = false;
    }
    h->analyzed = false;
    h->factored = false;
    std::string e = analyze_pattern(dim, colptr, rowval, index_base, h->sopts,
                                    h->opts.ordering == 2 ? h->user_perm.data() : nullptr, h->S);
    if (!e.empty()) return solver_set_error(h, OKKT_ERR_INVALID, e);
    h->analyze_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    h->analyzed = true;
    h->pat_colptr.assign(colptr, colptr + dim + 1);
    h->pat_rowval.assign(rowval, rowval + (colptr[dim] - colptr[0]));
    ++h->n_analyze_calls;
    return OKKT_OK;
  } catch (const std::bad_alloc&) {
    return solver_set_error(h, OKKT_ERR_ALLOC, "out of host memory in okkt_analyze");
  } catch (...) {
    return solver_set_error(h, OKKT_ERR_INTERNAL, "unexpected exception in okkt_analyze");
  }
}

int okkt_get_perm(okkt_handle h, int64_t* perm_out) {
  if (!h || !perm_out) return OKKT_ERR_INVALID;
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "not analysed");
  for (int64_t k = 0; k < h->S.n; ++k) perm_out[k] = h->S.perm[k];
  return OKKT_OK;
}

int okkt_get_etree(okkt_handle h, int64_t* parent_out, int64_t* colcount_out) {
  if (!h) return OKKT_ERR_INVALID;
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "not analysed");
  for (int64_t k = 0; k < h->S.n; ++k) {
    if (parent_out) parent_out[k] = h->S.parent[k];
    if (colcount_out) colcount_out[k] = h->S.colcount[k];
  }
  return OKKT_OK;
}

int okkt_get_stats(okkt_handle h, okkt_stats* out) {
  if (!h || !out) return OKKT_ERR_INVALID;
  std::memset(out, 0, sizeof(*out));
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "not analysed");
  const Symbolic& S = h->S;
  out->n = S.n;
  out->nnz_lower = S.nnz_lower;
  out->nnzL = S.nnzL;
  out->nnzL_stored = S.nnzL_stored;
  out->flops_exact = S.flops_exact;
  out->flops_stored = S.flops_stored;
  out->arena_bytes = (h->N.arena_doubles > 0 ? h->N.arena_doubles : S.arena_doubles) * 8;      // as allocated (panels + the shared contribution-block region) once the device plan exists
  out->nsuper = S.nsuper;
  out->nlevels = S.nlevels;
  out->max_front = S.max_front;
  int64_t nsmall = 0, nbig = 0;
  for (int s = 0; s < S.nsuper; ++s) {
    int64_t f = S.row_ptr[s + 1] - S.row_ptr[s];
    if (f <= std::max(32, std::min(h->sopts.small_front_max, 136))) ++nsmall; else ++nbig;
  }
  out->n_small_fronts = nsmall;
  out->n_big_fronts = nbig;
  out->sum_rowidx = (int64_t)S.rows.size();
  out->analyze_seconds = h->analyze_seconds;
  out->last_factor_ms = h->last_factor_ms;
  out->last_solve_ms = h->last_solve_ms;
  out->pattern_hash = S.pattern_hash;
  out->n_analyze_calls = h->n_analyze_calls;
  out->ordering_used = h->S.ordering_used;
  out->critical_pivots = h->S.critical_pivots;
  out->top_separator = h->S.top_separator;
  out->amd_skipped = h->S.amd_skipped ? 1 : 0;
  out->flops_other = h->S.flops_other;
  out->arena_dense_bytes = S.arena_doubles * 8;
  return OKKT_OK;
}

int okkt_factor_dev(okkt_handle h, const double* d_nzval, int64_t n, int64_t m, int sym_kind, okkt_inertia* out) {
  if (!h || (!d_nzval && h->S.nnz_in > 0)) return OKKT_ERR_INVALID;
  try {
    return solver_factor_device(h, d_nzval, n, m, sym_kind, out);
  } catch (...) {
    return solver_set_error(h, OKKT_ERR_INTERNAL, "unexpected exception in okkt_factor_dev");
  }
}

int okkt_factor(okkt_handle h, const double* nzval, int64_t n, int64_t m, int sym_kind, okkt_inertia* out) {
  if (!h || (!nzval && h->S.nnz_in > 0)) return OKKT_ERR_INVALID;
  try {
    int rc = solver_ensure_numeric(h);
    if (rc != OKKT_OK) return rc;
    if (h->S.nnz_in > 0) {
      hipError_t he = hipMemcpyAsync(h->N.vals_owned, nzval, (size_t)h->S.nnz_in * sizeof(double), hipMemcpyHostToDevice, h->stream);
      if (he != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, std::string("nzval upload: ") + hipGetErrorString(he));
    }
    return solver_factor_device(h, h->N.vals_owned, n, m, sym_kind, out);
  } catch (...) {
    return solver_set_error(h, OKKT_ERR_INTERNAL, "unexpected exception in okkt_factor");
  }
}

int okkt_solve_dev(okkt_handle h, const double* d_rhs, double* d_sol, int64_t nrhs) {
  if (!h || !d_rhs || !d_sol) return OKKT_ERR_INVALID;
  try {
    return solver_solve_device(h, d_rhs, d_sol, nrhs);
  } catch (...) {
    return solver_set_error(h, OKKT_ERR_INTERNAL, "unexpected exception in okkt_solve_dev");
  }
}

int okkt_solve(okkt_handle h, const double* rhs, double* sol, int64_t nrhs) {
  if (!h || !rhs || !sol) return OKKT_ERR_INVALID;
  try {
    int rc = solver_ensure_numeric(h);
    if (rc != OKKT_OK) return rc;
    if (!h->factored) return solver_set_error(h, OKKT_ERR_INVALID, "solve called before a factorisation");
    const int64_t len = h->S.n * std::max<int64_t>(nrhs, 0);
    if (len == 0) return OKKT_OK;
    if (h->rhs_stage_len < len) {
      if (h->d_rhs_stage) (void)hipFree(h->d_rhs_stage);
      h->d_rhs_stage = nullptr;
      h->rhs_stage_len = 0;
      if (hipMalloc((void**)&h->d_rhs_stage, (size_t)len * sizeof(double)) != hipSuccess)
        return solver_set_error(h, OKKT_ERR_ALLOC, "rhs staging allocation failed");
      h->rhs_stage_len = len;
    }
    hipError_t he = hipMemcpyAsync(h->d_rhs_stage, rhs, (size_t)len * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (he != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, std::string("rhs upload: ") + hipGetErrorString(he));
    rc = solver_solve_device(h, h->d_rhs_stage, h->d_rhs_stage, nrhs);
    if (rc != OKKT_OK) return rc;
    he = hipMemcpy(sol, h->d_rhs_stage, (size_t)len * sizeof(double), hipMemcpyDeviceToHost);
    if (he != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, std::string("sol download: ") + hipGetErrorString(he));
    return OKKT_OK;
  } catch (...) {
    return solver_set_error(h, OKKT_ERR_INTERNAL, "unexpected exception in okkt_solve");
  }
}

int okkt_get_diag(okkt_handle h, double* d_out) {
  if (!h || !d_out) return OKKT_ERR_INVALID;
  int rc = solver_ensure_numeric(h);
  if (rc != OKKT_OK) return rc;
  if (!h->factored) return solver_set_error(h, OKKT_ERR_INVALID, "no factorisation");
  if (hipMemcpy(d_out, h->N.d.dvals, (size_t)h->S.n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return solver_set_error(h, OKKT_ERR_HIP, "download of D failed");
  return OKKT_OK;
}

int okkt_get_factor_csc(okkt_handle h, int64_t* colptr_out, int64_t* rowval_out, double* val_out, int64_t* nnz_out) {
  if (!h) return OKKT_ERR_INVALID;
  if (!h->analyzed) return solver_set_error(h, OKKT_ERR_INVALID, "not analysed");
  const Symbolic& S = h->S;
  const int64_t nnz = S.nnzL_stored - S.n;
  if (nnz_out) *nnz_out = nnz;
  if (!colptr_out || !rowval_out || !val_out) return OKKT_OK;
  try {
    int rc = solver_ensure_numeric(h);
    if (rc != OKKT_OK) return rc;
    if (!h->factored) return solver_set_error(h, OKKT_ERR_INVALID, "no factorisation");
    std::vector<double> front;
    int64_t q = 0;
    for (int s = 0; s < S.nsuper; ++s) {
      const int64_t f = S.row_ptr[s + 1] - S.row_ptr[s];
      const int64_t k = S.sn_col0[s + 1] - S.sn_col0[s];
      front.resize((size_t)(f * k));
      if (hipMemcpy(front.data(), h->N.d.arena + h->N.front_pos_host[s], (size_t)(f * k) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return solver_set_error(h, OKKT_ERR_HIP, "download of a front failed");
      for (int64_t lc = 0; lc < k; ++lc) {
        colptr_out[S.sn_col0[s] + lc] = q;
        for (int64_t i = lc + 1; i < f; ++i) {
          rowval_out[q] = S.rows[S.row_ptr[s] + i];
          val_out[q] = front[(size_t)(lc * f + i)];
          ++q;
        }
      }
    }
    colptr_out[S.n] = q;
    return OKKT_OK;
  } catch (...) {
    return solver_set_error(h, OKKT_ERR_INTERNAL, "unexpected exception in okkt_get_factor_csc");
  }
}

int okkt_dev_alloc(okkt_handle h, int64_t bytes, void** out) {
  if (!h || !out || bytes < 0) return OKKT_ERR_INVALID;
  int rc = ensure_device(h);
  if (rc != OKKT_OK) return rc;
  if (hipMalloc(out, (size_t)std::max<int64_t>(bytes, 8)) != hipSuccess) return solver_set_error(h, OKKT_ERR_ALLOC, "hipMalloc failed");
  return OKKT_OK;
}
int okkt_dev_free(okkt_handle h, void* p) {
  if (!h) return OKKT_ERR_INVALID;
  int rc = ensure_device(h);
  if (rc != OKKT_OK) return rc;
  (void)hipStreamSynchronize(h->stream);
  return hipFree(p) == hipSuccess ? OKKT_OK : solver_set_error(h, OKKT_ERR_HIP, "hipFree failed");
}
int okkt_dev_upload(okkt_handle h, void* d_dst, const void* src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!d_dst || !src))) return OKKT_ERR_INVALID;
  int rc = ensure_device(h);
  if (rc != OKKT_OK) return rc;
  if (bytes == 0) return OKKT_OK;
  (void)hipStreamSynchronize(h->stream);
  return hipMemcpy(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice) == hipSuccess ? OKKT_OK : solver_set_error(h, OKKT_ERR_HIP, "upload failed");
}
int okkt_dev_download(okkt_handle h, void* dst, const void* d_src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !d_src))) return OKKT_ERR_INVALID;
  int rc = ensure_device(h);
  if (rc != OKKT_OK) return rc;
  if (bytes == 0) return OKKT_OK;
  (void)hipStreamSynchronize(h->stream);
  return hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost) == hipSuccess ? OKKT_OK : solver_set_error(h, OKKT_ERR_HIP, "download failed");
}
void* okkt_get_stream(okkt_handle h) { return h ? (void*)h->stream : nullptr; }

int okkt_profile_dominant(okkt_handle h, int enable) {
  if (!h) return OKKT_ERR_INVALID;
  int rc = ensure_device(h);
  if (rc != OKKT_OK) return rc;
  (void)hipStreamSynchronize(h->stream);
  h->N.profile = enable != 0;
  h->N.prof_used = 0;
  h->N.prof_flops.clear();
  return OKKT_OK;
}

int okkt_get_profile(okkt_handle h, int64_t* n_launches, double* total_ms, double* total_flops) {
  if (!h || !n_launches || !total_ms || !total_flops) return OKKT_ERR_INVALID;
  int rc = ensure_device(h);
  if (rc != OKKT_OK) return rc;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return solver_set_error(h, OKKT_ERR_HIP, "stream sync failed");
  double ms = 0, fl = 0;
  const size_t n = h->N.prof_used / 2;
  for (size_t i = 0; i < n; ++i) {
    float t = 0;
    if (hipEventElapsedTime(&t, h->N.prof_events[2 * i], h->N.prof_events[2 * i + 1]) != hipSuccess)
      return solver_set_error(h, OKKT_ERR_HIP, "hipEventElapsedTime failed");
    ms += t;
    fl += h->N.prof_flops[i];
    if (getenv("OKKT_DEBUG_SYRK_LOG")) fprintf(stderr, "syrk launch %3zu: %9.1f us %8.3f GFLOP %6.1f TFLOP/s\n", i, t * 1e3, h->N.prof_flops[i] * 1e-9, h->N.prof_flops[i] / (t * 1e-3) * 1e-12);
  }
  if (getenv("OKKT_DEBUG_SYRK_LOG") && h->N.d.zero_page) {   // phase ticks of k_big_syrk<16, .> (OKKT_DEBUG_SYRK=96), 10 ns each
    unsigned long long T[8];
    if (hipMemcpy(T, h->N.d.zero_page + 256, sizeof(T), hipMemcpyDeviceToHost) == hipSuccess && T[0] > 0) {
      fprintf(stderr, "syrk workgroups %llu: per workgroup (us) start->first chunk ready %.2f, main loop %.2f (%.3f per 16-column chunk), store issue %.2f, store drain %.2f\n",
              T[0], T[1] * 0.01 / T[0], T[2] * 0.01 / T[0], T[2] * 0.01 / (double)T[5], T[3] * 0.01 / T[0], T[4] * 0.01 / T[0]);
      fprintf(stderr, "   of the start: scalar set-up %.2f us, C tile loaded (if waited for) %.2f us\n", T[6] * 0.01 / T[0], T[7] * 0.01 / T[0]);
      (void)hipMemset(h->N.d.zero_page + 256, 0, sizeof(T));
    }
  }
  *n_launches = (int64_t)n;
  *total_ms = ms;
  *total_flops = fl;
  return OKKT_OK;
}

int64_t okkt_debug_dataflow_queue(int32_t nfronts, const int32_t* f, const int32_t* k, int32_t workers, int32_t group,
                                  int32_t* tasks, int64_t cap, double* model_us) {
  if (nfronts < 0 || !f || !k || (cap > 0 && !tasks)) return OKKT_ERR_INVALID;
  try {
    std::vector<okkt::DfFront> fronts;
    for (int a = 0; a < nfronts; ++a) {
      if (k[a] < 1 || f[a] < k[a]) return OKKT_ERR_INVALID;
      fronts.push_back({a, f[a], k[a]});
    }
    std::vector<okkt::DfTask> q;
    double model = 0;
    okkt::df_build_queue(fronts, workers, group & 255, std::max(1, (group >> 8) & 255), (group >> 16) & 1, (group >> 17) & 1, q, &model, (group >> 18) & 1, (group >> 19) & 1);
    if (model_us) *model_us = model;
    for (int64_t t = 0; t < (int64_t)q.size() && t < cap; ++t) {
      tasks[4 * t] = q[t].front; tasks[4 * t + 1] = q[t].type_nq; tasks[4 * t + 2] = q[t].ij; tasks[4 * t + 3] = q[t].q0;
    }
    return (int64_t)q.size();
  } catch (...) { return OKKT_ERR_ALLOC; }
}

}  // extern "C"
