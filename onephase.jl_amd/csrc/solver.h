// Internal C++ object behind okkt_handle.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/okkt.h"
#include "numeric.h"
#include "symbolic.h"

struct okkt_solver_s {
  okkt_opts opts;
  okkt::SymbolicOptions sopts;
  okkt::Symbolic S;
  okkt::Numeric N;
  bool analyzed = false;
  bool factored = false;
  bool device_ready = false;  // HIP device selected and stream created
  bool numeric_ready = false; // device plan uploaded for the current pattern
  bool last_failed = false;   // the last factorisation did not give the wanted inertia (the next one is a retry of the delta loop)
  bool early_exit = false;    // stop a factorisation whose inertia is already wrong (set by the KKT level for factor! / the delta loop)
  int device = 0;
  hipStream_t stream = nullptr;        // the handle's stream (all CUs)
  hipStream_t stream_masked = nullptr; // look-ahead main stream: CU mask without the reserved CUs (segments that use the look-ahead run here)
  int stream_la = 0, stream_reserved = 0;   // key of the pooled stream set (api.cpp)
  int stream_seq = 0;                       // its order of creation in the process
  hipStream_t stream_panel = nullptr;  // look-ahead panel stream (high priority, all CUs)
  hipStream_t stream_aux = nullptr;    // second panel stream: the part of the in-group updates that k_big_diag does not wait for
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<int64_t> user_perm;
  std::vector<int64_t> pat_colptr, pat_rowval;   // the analysed pattern as the caller passed it (exact re-use test in okkt_analyze)
  std::string err;
  double analyze_seconds = 0, last_factor_ms = 0, last_solve_ms = 0;
  int64_t n_analyze_calls = 0;
  int part_id = 0;                 // multi-GPU part of this handle (okkt_dist_set_partition)
  const double* dist_vals = nullptr;
  int64_t dist_n = 0, dist_m = 0;
  int dist_kind = 0;
  double dist_tol = 0;
  // RCCL transport of the sharded path (dist.cpp): communicator and the three exchange buffers
  void* rccl_comm = nullptr;
  int rccl_nranks = 0, rccl_rank = 0;
  double *dist_cb = nullptr, *dist_cv = nullptr, *dist_x = nullptr;
  size_t dist_cb_cap = 0, dist_cv_cap = 0, dist_x_cap = 0;   // doubles the exchange buffers were allocated for (okkt_dist_comm_init)
  long long* dist_counts = nullptr;  // 4 summed pivot counts on the device
  double* d_rhs_stage = nullptr;  // staging for host-side rhs/sol
  int64_t rhs_stage_len = 0;
};

namespace okkt {
// shared by the linear-solver level and the KKT level
int solver_factor_device(okkt_solver_s* h, const double* d_vals, int64_t n, int64_t m, int sym_kind,
                         okkt_inertia* out);
int solver_solve_device(okkt_solver_s* h, const double* d_rhs, double* d_sol, int64_t nrhs);
// the same without synchronisation or timing (the KKT level strings solves and vector kernels together on the stream);
// accumulate: d_sol += F \ d_rhs
int solver_solve_enqueue(okkt_solver_s* h, const double* d_rhs, double* d_sol, int64_t nrhs, bool accumulate);
int solver_set_error(okkt_solver_s* h, int code, const std::string& msg);
int solver_ensure_numeric(okkt_solver_s* h);
}  // namespace okkt
