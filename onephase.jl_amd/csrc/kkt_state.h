// State of a KKT-system handle (include/okkt.h, level 2), shared by kkt.hip (system, factor, direction) and
// linesearch.hip (step-side vector kernels).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "solver.h"

struct okkt_kkt_s {
  okkt_handle ls = nullptr;
  int kind = OKKT_KKT_SCHUR;
  int64_t n = 0, m = 0, nnzH = 0, nnzJ = 0, nnzA = 0, dimA = 0;
  bool structured = false, formed = false, factored = false;
  bool have_dxnorm = false;   // dxnorm = norm(dir.x, Inf) of the resident direction is known (step-side functions)
  double dxnorm = 0.0;
  bool have_dir = false;   // dx, dy, ds hold the direction of the last okkt_kkt_compute_direction for the current (s, y)
  double delta = 0.0;
  std::string err;
  std::vector<void*> allocs;
  // assembled matrix pattern (host copy, 0-based CSC lower)
  std::vector<int64_t> Ap, Ai;
  // device: H (CSC lower + CSR view), J (CSC + CSR view), point
  int64_t *Hp = nullptr, *Hrp = nullptr, *Hrmap = nullptr, *Jp = nullptr, *Jrp = nullptr, *Jrmap = nullptr;
  int *Hi = nullptr, *Hrj = nullptr, *Ji = nullptr, *Jrj = nullptr;
  double *Hx = nullptr, *Jx = nullptr, *s = nullptr, *y = nullptr, *sig = nullptr;
  double* Avals = nullptr;
  int64_t *mapH = nullptr, *mapJ = nullptr, *diagA = nullptr;   // symmetric: value slots in A
  int64_t *qptr = nullptr, *qh = nullptr;                        // schur: contributions per Q entry
  int *qa = nullptr, *qb = nullptr, *qi = nullptr;
  // schur, LDS-staged assembly: per CSC entry (row i, column b) of J the start of row i's CSR segment with columns >= b and
  // its first term; per term the slot of its Q entry inside column b (16 bits); groups of 16 lanes own one column of Q
  int64_t *seg_q = nullptr, *seg_t = nullptr, *dAp64 = nullptr;
  uint16_t* tslot = nullptr;
  int schur_groups = 0, schur_maxcol = 0;
  double* schur_diag = nullptr;
  // work vectors
  double *rD = nullptr, *rP = nullptr, *rC = nullptr, *dx = nullptr, *dy = nullptr, *ds = nullptr;
  double *vn1 = nullptr, *vn2 = nullptr, *vn3 = nullptr, *vm1 = nullptr, *vm2 = nullptr, *big1 = nullptr, *big2 = nullptr;
  double* red = nullptr;  // reduction outputs
  // batched directions (okkt_kkt_compute_directions): gradient and constraint values of the last okkt_kkt_system_rhs kept resident,
  // per-rhs slots of the rhs triples, the linear-system rhs / solution, the directions and the error reductions
  double *cur_grad = nullptr, *cur_cons = nullptr;
  double cur_mu = 0.0, cur_pen = 0.0;
  int batch_cap = 0;
  double *b_rD = nullptr, *b_rP = nullptr, *b_rC = nullptr, *b_rhs = nullptr, *b_sol = nullptr, *b_res = nullptr, *b_dx = nullptr, *b_dy = nullptr, *b_ds = nullptr, *b_red = nullptr;
  double* ones = nullptr; // max(n, m) ones (row / column sums of the diagonal-dominance scan), allocated on first use
  int diag_dom_warnings = 0;   // failed attempts of the last ipopt_strategy! whose x-block was diagonally dominant (delta_strategy.jl:95)
  // step-side kernels (linesearch.hip): staged host vectors and reduction partials, allocated on first use
  double *ls_m[4] = {nullptr, nullptr, nullptr, nullptr}, *ls_n[2] = {nullptr, nullptr}, *ls_part = nullptr, *ls_out = nullptr;
  double* Jcur = nullptr; // Jacobian values of a current iterate that differs from the factorised one
  // CSR-ordered copies of the values (refreshed by form_system): the row-wise products read values and column indices
  // contiguously instead of gathering through the CSC -> CSR map
  double *Jcsr = nullptr, *Hcsr = nullptr, *Hdiag = nullptr, *Jcur_csr = nullptr;
  int lprJr = 16, lprJc = 16, lprH = 8;      // lanes per row / column of the segmented products (from the average lengths)
  // current iterate of the last okkt_kkt_system_rhs (kkt_associate_rhs!, schur.jl:34-45): Schur_KKT_solver_direct reads it
  double *cur_s = nullptr, *cur_y = nullptr, *cur_sig = nullptr;
  const double* cur_Jx = nullptr;            // CSC values of the current iterate's Jacobian (Jx or Jcur)
  const double* cur_Jcsr = nullptr;          // the same in CSR order
  bool have_cur = false, have_rhs = false;
  double* part = nullptr;                    // per-workgroup partial maxima of the N-err kernels
  int64_t part_blocks = 0;
  // device timers: (tag, start, stop) event segments of the last call of each kind, summed per tag on request
  struct Timer {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    struct Seg { int tag; size_t a, b; };
    std::vector<Seg> segs;
    void reset() { used = 0; segs.clear(); }
    size_t mark(hipStream_t st) {
      if (used == ev.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return used; ev.push_back(e); }
      (void)hipEventRecord(ev[used], st);
      return used++;
    }
    void seg(int tag, size_t a, size_t b) { if (a < used && b < used) segs.push_back(Seg{tag, a, b}); }
    double sum(int tag) const {
      double tot = 0.0;
      for (const Seg& g : segs) { float ms = 0; if (g.tag == tag && hipEventElapsedTime(&ms, ev[g.a], ev[g.b]) == hipSuccess) tot += ms; }
      return tot;
    }
    void destroy() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); ev.clear(); reset(); }
  };
  Timer tm_form, tm_factor, tm_rhs, tm_dir;
  double t_factor_ms = 0.0;
  int n_solves = 0;
  // ---- clever symmetric (clever_symmetric.jl): parallel-row groups and the reduced system
  bool indexed = false;
  int64_t m_new = 0;
  int rescale_mode = OKKT_RESCALE_NONE;
  double rescale_mu = 0.0, rescale_xinf = 0.0;
  std::vector<int64_t> h_Jrp, h_Jrmap;           // host CSR view of J (kept for compute_indicies)
  std::vector<int> h_Jrj;
  std::vector<int64_t> h_Hp, h_Jp;               // host copies of the column pointers / row indices
  std::vector<int> h_Hi, h_Ji;
  std::vector<int64_t> h_first, h_gptr, h_mind;  // groups: first row, member ranges, member rows (ls order)
  std::vector<double> h_mratio;
  int64_t *gptr = nullptr, *mapJc = nullptr, *Arp = nullptr, *Armap = nullptr, *dAp = nullptr;
  int *mind = nullptr, *row_grp = nullptr, *dAi = nullptr, *Arj = nullptr, *Hcol = nullptr, *Jcol = nullptr;
  double *mratio = nullptr, *row_ratio = nullptr, *gU = nullptr, *rowg = nullptr, *Dres = nullptr, *true_x_diag = nullptr;
  double *crhs = nullptr, *big3 = nullptr, *big4 = nullptr;
};

#define KK_TRY(k, expr)                                                                      \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) { (k)->err = std::string(#expr) + ": " + hipGetErrorString(e__); return OKKT_ERR_HIP; } \
  } while (0)

namespace okkt {
// y = J x (m), y = J' v with the Jacobian values Jx (n), y = H x with the lower-stored H (n): on the handle's stream
void kk_spmv_J(okkt_kkt_s* k, const double* x, double* y);
void kk_spmv_JT(okkt_kkt_s* k, const double* Jx, const double* v, double* y);
void kk_spmv_H(okkt_kkt_s* k, const double* x, double* y);
inline hipStream_t kk_stream(okkt_kkt_s* k) { return k->ls->stream; }
}  // namespace okkt
