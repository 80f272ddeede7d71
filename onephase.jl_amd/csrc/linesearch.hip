// C ABI, step-side vector kernels (include/okkt.h): what simple_ls computes either side of a
// primal-dual step from the cached J, H, the point and the direction -- on the device, against the state the KKT
// handle already holds.
//
//   lb_s_thres / lb_s_predict / lb_y / lb_s / simple_max_step   /root/reference/src/line_search/frac_boundary.jl:3-35
//   the s-bound test of move_primal, dual_bounds, move_dual      src/line_search/move.jl:15-17,28-80,82-118
//   how simple_ls combines them                                   src/line_search/line_search.jl:40-41,84-86
//   comp, eval_grad_phi, phi_predicted_reduction_primal_dual,
//   comp_predicted, merit_function_predicted_reduction            src/utils/eval.jl:11-13,117-120,236-273
//
// All of it is HBM-bound streaming: one pass over a few length-m (or length-n) vectors, a map per entry and a
// reduction.  A launch is (up to) 256 workgroups of 256 threads, grid-stride over the entries, one partial per
// channel and workgroup, then a one-workgroup final pass: fixed partition, so sums are reproducible; max / min
// propagate NaN the way Julia's maximum / max / min do.  This file is compiled with -ffp-contract=off: the maps
// are written operation by operation as the reference evaluates them, and the max / min results are bit-exact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>

#include "kkt_state.h"

using namespace okkt;

namespace {

enum { kSum = 0, kNanMax = 1, kNanMin = 2 };
constexpr int kMaxBlocks = 256;

template <int NCH>
struct Channels { int op[NCH]; double init[NCH]; };

__device__ __forceinline__ double combine(int op, double a, double b) {
  if (op == kSum) return a + b;
  if (a != a || b != b) return NAN;
  return op == kNanMax ? fmax(a, b) : fmin(a, b);
}
__device__ __forceinline__ double nan_min(double a, double b) { return (a != a || b != b) ? NAN : fmin(a, b); }

template <int NCH>
__device__ void block_reduce(double (&v)[NCH], const Channels<NCH>& ch, double* out) {
  __shared__ double sh[NCH][256];
  const int t = threadIdx.x;
#pragma unroll
  for (int c = 0; c < NCH; ++c) sh[c][t] = v[c];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) sh[c][t] = combine(ch.op[c], sh[c][t], sh[c][t + o]);
    }
    __syncthreads();
  }
  if (t == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) out[c] = sh[c][0];
  }
}

// partials: part[block * NCH + c]
template <int NCH, typename F>
__global__ __launch_bounds__(256) void k_map_reduce(int64_t n, F f, Channels<NCH> ch, double* __restrict__ part) {
  double acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = ch.op[c] == kSum ? 0.0 : ch.init[c];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    double v[NCH];
    f(i, v);
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = combine(ch.op[c], acc[c], v[c]);
  }
  block_reduce<NCH>(acc, ch, part + (size_t)blockIdx.x * NCH);
}
template <int NCH>
__global__ __launch_bounds__(256) void k_reduce_final(int nb, Channels<NCH> ch, const double* __restrict__ part, double* __restrict__ out) {
  double acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = ch.op[c] == kSum ? 0.0 : ch.init[c];
  for (int b = threadIdx.x; b < nb; b += 256) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = combine(ch.op[c], acc[c], part[(size_t)b * NCH + c]);
  }
  block_reduce<NCH>(acc, ch, out);
}

// ---- the maps ------------------------------------------------------------------------------------------------
struct MapAbs {   // norm(v, Inf)
  const double* v;
  __device__ void operator()(int64_t i, double (&o)[1]) const { o[0] = fabs(v[i]); }
};
// -dir ./ (val - frac .* min.(s, thr)): simple_max_step with lb = frac .* lb_s_thres (frac_boundary.jl:3-15,31-35)
struct MapRatioS {
  const double *val, *dir, *frac, *s;
  double thr;
  __device__ void operator()(int64_t i, double (&o)[1]) const {
    const double lb = frac[i] * nan_min(s[i], thr);
    o[0] = -dir[i] / (val[i] - lb);
  }
};
// s_new .>= frac .* min.(s, thr) (move.jl:15; frac_boundary.jl:22-28): 1 / 0, reduced with min
struct MapSBound {
  const double *s_new, *frac, *s;
  double thr;
  __device__ void operator()(int64_t i, double (&o)[1]) const { o[0] = s_new[i] >= frac[i] * nan_min(s[i], thr) ? 1.0 : 0.0; }
};
// dual_bounds (move.jl:28-80) entry i as (candidate for lb, candidate for ub, index if it resets the interval),
// counted only behind the last reset `after`; and -dy ./ (y_cand - frac .* y * min(1, |dx|_inf)) (line_search.jl:85-86)
struct MapDualBounds {
  const double *s_c, *y_c, *dy, *frac, *y;
  double mu, comp_feas, nxmin;
  int64_t after;
  __device__ void operator()(int64_t i, double (&o)[4]) const {
    const double s = s_c[i], d = dy[i], yc = y_c[i];
    const double safety_factor = 1.001, safety_add = 0.0;
    const double ub_dyi = mu / (comp_feas * s * d) - yc / d;
    const double lb_dyi = mu * comp_feas / (s * d) - yc / d;
    double a = -INFINITY, b = INFINITY, r = -1.0;
    if (d > 0.0) { a = lb_dyi * safety_factor + safety_add; b = ub_dyi / safety_factor - safety_add; }
    else if (d < 0.0) { a = ub_dyi * safety_factor + safety_add; b = lb_dyi / safety_factor - safety_add; }
    else if (lb_dyi >= 0.0 || ub_dyi <= 0.0) r = (double)i;
    if (i <= after) { a = -INFINITY; b = INFINITY; }
    o[0] = a; o[1] = b; o[2] = r;
    o[3] = -d / (yc - frac[i] * y[i] * nxmin);
  }
};
// m-part of the predicted reductions (eval.jl:236-273): (J dx).^2 . (y ./ s), |comp|, |comp_predicted|
struct MapPredM {
  const double *v, *s, *y, *dy, *ds;
  double mu, mu_new, step;
  __device__ void operator()(int64_t i, double (&o)[3]) const {
    const double si = s[i], yi = y[i];
    o[0] = v[i] * v[i] * (yi / si);
    o[1] = fabs(si * yi - mu);
    o[2] = fabs(si * yi + dy[i] * si * step + ds[i] * yi * step - mu_new);
  }
};
// n-part: dx . (grad - J'w) and dx . (H dx)
struct MapPredN {
  const double *dx, *grad, *jtw, *h;
  __device__ void operator()(int64_t i, double (&o)[2]) const {
    o[0] = dx[i] * (grad[i] - jtw[i]);
    o[1] = dx[i] * h[i];
  }
};
// move_dual (move.jl:100-112): res . q and q . q, first n entries / last m entries
struct MapDualStepN {
  const double *grad, *jtw, *jtdy;
  double scale_D;
  __device__ void operator()(int64_t i, double (&o)[2]) const {
    const double q = scale_D * jtdy[i], res = scale_D * (grad[i] - jtw[i]);
    o[0] = res * q; o[1] = q * q;
  }
};
struct MapDualStepM {
  const double *s_c, *y_c, *dy;
  double scale_mu, mu;
  __device__ void operator()(int64_t i, double (&o)[2]) const {
    const double q = scale_mu * s_c[i] * dy[i], res = -scale_mu * (s_c[i] * y_c[i] - mu);
    o[0] = res * q; o[1] = q * q;
  }
};
__global__ void k_axpb(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ o, int recip) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = (recip ? a / x[i] : a * x[i]) + b;   // recip: a ./ x + b
}

int ls_fail(okkt_kkt_s* k, int code, const char* msg) { k->err = msg; return code; }

int ls_ready(okkt_kkt_s* k, bool need_dir) {
  if (!k) return OKKT_ERR_INVALID;
  if (!k->formed) return ls_fail(k, OKKT_ERR_INVALID, "okkt_kkt_form_system has not been called");
  if (need_dir && !k->have_dir) return ls_fail(k, OKKT_ERR_INVALID, "no direction: call okkt_kkt_compute_direction for this system first");
  KK_TRY(k, hipSetDevice(k->ls->device));
  if (!k->ls_part) {
    auto grab = [&](size_t count, double** out) -> int {
      void* p = nullptr;
      KK_TRY(k, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(double)));
      k->allocs.push_back(p);
      *out = (double*)p;
      return OKKT_OK;
    };
    int rc;
    for (int q = 0; q < 4; ++q) if ((rc = grab((size_t)k->m, &k->ls_m[q])) != OKKT_OK) return rc;
    for (int q = 0; q < 2; ++q) if ((rc = grab((size_t)k->n, &k->ls_n[q])) != OKKT_OK) return rc;
    if ((rc = grab(8, &k->ls_out)) != OKKT_OK) return rc;
    if ((rc = grab((size_t)kMaxBlocks * 8, &k->ls_part)) != OKKT_OK) return rc;
  }
  return OKKT_OK;
}

int stage(okkt_kkt_s* k, double* dst, const double* host, int64_t count) {
  if (count > 0) KK_TRY(k, hipMemcpyAsync(dst, host, (size_t)count * 8, hipMemcpyHostToDevice, kk_stream(k)));
  return OKKT_OK;
}

// map + reduce over n entries into host_out[NCH]; n == 0 gives the identities
template <int NCH, typename F>
int map_reduce(okkt_kkt_s* k, int64_t n, const F& f, const Channels<NCH>& ch, double* host_out) {
  if (n <= 0) {
    for (int c = 0; c < NCH; ++c) host_out[c] = ch.op[c] == kSum ? 0.0 : ch.init[c];
    return OKKT_OK;
  }
  hipStream_t st = kk_stream(k);
  const int nb = (int)std::min<int64_t>(kMaxBlocks, (n + 1023) / 1024);
  hipLaunchKernelGGL((k_map_reduce<NCH, F>), dim3(nb), dim3(256), 0, st, n, f, ch, k->ls_part);
  hipLaunchKernelGGL((k_reduce_final<NCH>), dim3(1), dim3(256), 0, st, nb, ch, k->ls_part, k->ls_out);
  KK_TRY(k, hipMemcpyAsync(host_out, k->ls_out, NCH * sizeof(double), hipMemcpyDeviceToHost, st));
  KK_TRY(k, hipStreamSynchronize(st));
  KK_TRY(k, hipGetLastError());
  return OKKT_OK;
}

// norm(dir.x, Inf) and lb_s_thres's scalar  norm * norm^ex  (frac_boundary.jl:3-10); pow on the host (one libm for all)
int dx_norm(okkt_kkt_s* k, double* nx) {
  if (k->have_dxnorm) { *nx = k->dxnorm; return OKKT_OK; }     // once per direction
  Channels<1> ch{{kNanMax}, {0.0}};
  const int rc = map_reduce<1>(k, k->n, MapAbs{k->dx}, ch, nx);
  if (rc == OKKT_OK) { k->dxnorm = *nx; k->have_dxnorm = true; }
  return rc;
}
double thres_scalar(double nx, double ex) { return nx * std::pow(nx, ex); }
double jl_min(double a, double b) { return (a != a || b != b) ? NAN : std::min(a, b); }
double jl_max(double a, double b) { return (a != a || b != b) ? NAN : std::max(a, b); }

}  // namespace

extern "C" {

int okkt_kkt_set_direction(okkt_kkt_handle k, const double* dx, const double* dy, const double* ds) {
  int rc = ls_ready(k, false);
  if (rc != OKKT_OK) return rc;
  if ((k->n > 0 && !dx) || (k->m > 0 && (!dy || !ds))) return OKKT_ERR_INVALID;
  if ((rc = stage(k, k->dx, dx, k->n)) != OKKT_OK || (rc = stage(k, k->dy, dy, k->m)) != OKKT_OK || (rc = stage(k, k->ds, ds, k->m)) != OKKT_OK) return rc;
  KK_TRY(k, hipStreamSynchronize(kk_stream(k)));
  k->have_dir = true;
  k->have_dxnorm = false;
  return OKKT_OK;
}

int okkt_kkt_max_step_primal(okkt_kkt_handle k, const double* frac_bd_predict, double ex, double* step_size_P, double* dx_norm_inf) {
  int rc = ls_ready(k, true);
  if (rc != OKKT_OK) return rc;
  if (!step_size_P || (k->m > 0 && !frac_bd_predict)) return OKKT_ERR_INVALID;
  double nx = 0.0;
  if ((rc = dx_norm(k, &nx)) != OKKT_OK) return rc;
  if ((rc = stage(k, k->ls_m[0], frac_bd_predict, k->m)) != OKKT_OK) return rc;
  double ratio = 1.0;
  Channels<1> ch{{kNanMax}, {1.0}};
  if ((rc = map_reduce<1>(k, k->m, MapRatioS{k->s, k->ds, k->ls_m[0], k->s, thres_scalar(nx, ex)}, ch, &ratio)) != OKKT_OK) return rc;
  *step_size_P = 1.0 / ratio;
  if (dx_norm_inf) *dx_norm_inf = nx;
  return OKKT_OK;
}

int okkt_kkt_s_bound_ok(okkt_kkt_handle k, const double* s_new, const double* frac_bd, double ex, int32_t* ok) {
  int rc = ls_ready(k, true);
  if (rc != OKKT_OK) return rc;
  if (!ok || (k->m > 0 && (!s_new || !frac_bd))) return OKKT_ERR_INVALID;
  double nx = 0.0;
  if ((rc = dx_norm(k, &nx)) != OKKT_OK) return rc;
  if ((rc = stage(k, k->ls_m[0], frac_bd, k->m)) != OKKT_OK || (rc = stage(k, k->ls_m[1], s_new, k->m)) != OKKT_OK) return rc;
  double all_ok = 1.0;
  Channels<1> ch{{kNanMin}, {1.0}};
  if ((rc = map_reduce<1>(k, k->m, MapSBound{k->ls_m[1], k->ls_m[0], k->s, thres_scalar(nx, ex)}, ch, &all_ok)) != OKKT_OK) return rc;
  *ok = all_ok == 1.0 ? 1 : 0;
  return OKKT_OK;
}

int okkt_kkt_dual_step_range(okkt_kkt_handle k, const double* s_cand, const double* y_cand, double mu_cand, double comp_feas,
                             const double* frac_bd, double* lb_out, double* ub_out) {
  int rc = ls_ready(k, true);
  if (rc != OKKT_OK) return rc;
  if (!lb_out || !ub_out || (k->m > 0 && (!s_cand || !y_cand || !frac_bd))) return OKKT_ERR_INVALID;
  double nx = 0.0;
  if ((rc = dx_norm(k, &nx)) != OKKT_OK) return rc;
  if ((rc = stage(k, k->ls_m[0], frac_bd, k->m)) != OKKT_OK || (rc = stage(k, k->ls_m[1], s_cand, k->m)) != OKKT_OK ||
      (rc = stage(k, k->ls_m[2], y_cand, k->m)) != OKKT_OK)
    return rc;
  // lb = max over the entries behind the last reset (start 0), ub = min (start 1, or -1 behind a reset), the last
  // reset index, and the ratio of simple_max_step(y_cand, dy, lb_y)
  MapDualBounds f{k->ls_m[1], k->ls_m[2], k->dy, k->ls_m[0], k->y, mu_cand, comp_feas, jl_min(1.0, nx), -1};
  Channels<4> ch{{kNanMax, kNanMin, kNanMax, kNanMax}, {0.0, 1.0, -1.0, 1.0}};
  double r[4];
  if ((rc = map_reduce<4>(k, k->m, f, ch, r)) != OKKT_OK) return rc;
  if (r[2] >= 0.0) {   // an entry with dy == 0 reset the interval to (0, -1): only what follows it counts
    f.after = (int64_t)r[2];
    ch.init[1] = -1.0;
    const double ratio_y = r[3];
    if ((rc = map_reduce<4>(k, k->m, f, ch, r)) != OKKT_OK) return rc;
    r[3] = ratio_y;
  }
  double lb = r[0], ub = r[1];
  if (!std::isfinite(lb) || !std::isfinite(ub)) { lb = 0.0; ub = -1.0; }   // isbad(lb) || isbad(ub), move.jl:74-76
  ub = jl_min(ub, 1.0 / r[3]);
  *lb_out = lb; *ub_out = ub;
  return OKKT_OK;
}

int okkt_kkt_predicted_reduction(okkt_kkt_handle k, const double* grad, double mu, double dmu, double a_norm_penalty,
                                 double step_size, double out[4]) {
  int rc = ls_ready(k, true);
  if (rc != OKKT_OK) return rc;
  if (!out || (k->n > 0 && !grad)) return OKKT_ERR_INVALID;
  hipStream_t st = kk_stream(k);
  const int64_t n = k->n, m = k->m;
  if ((rc = stage(k, k->ls_n[0], grad, n)) != OKKT_OK) return rc;
  double rm[3] = {0.0, 0.0, 0.0}, rn[2] = {0.0, 0.0};
  if (m) {
    kk_spmv_J(k, k->dx, k->ls_m[0]);                                                          // v = J dx
    // eval_grad_phi = grad - J'(mu ./ s) + mu * pen * J'1 = grad - J'(mu ./ s - mu * pen)
    hipLaunchKernelGGL(k_axpb, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, mu, k->s, -(mu * a_norm_penalty), k->ls_m[1], 1);
    Channels<3> ch{{kSum, kNanMax, kNanMax}, {0.0, 0.0, 0.0}};
    if ((rc = map_reduce<3>(k, m, MapPredM{k->ls_m[0], k->s, k->y, k->dy, k->ds, mu, mu + dmu * step_size, step_size}, ch, rm)) != OKKT_OK) return rc;
  }
  if (n) {
    if (m) kk_spmv_JT(k, k->Jx, k->ls_m[1], k->ls_n[1]);
    else KK_TRY(k, hipMemsetAsync(k->ls_n[1], 0, (size_t)n * 8, st));
    kk_spmv_H(k, k->dx, k->vn1);
    Channels<2> ch{{kSum, kSum}, {0.0, 0.0}};
    if ((rc = map_reduce<2>(k, n, MapPredN{k->dx, k->ls_n[0], k->ls_n[1], k->vn1}, ch, rn)) != OKKT_OK) return rc;
  }
  const double phi_red = step_size * rn[0] + step_size * step_size * 0.5 * (rn[1] + rm[0]);
  const double C_k = rm[1], P_k = rm[2];
  const double comp_penalty = m > 0 ? (P_k * P_k * P_k - C_k * C_k * C_k) / (mu * mu) : 0.0;
  out[0] = phi_red; out[1] = C_k; out[2] = P_k; out[3] = phi_red + comp_penalty;
  return OKKT_OK;
}

int okkt_kkt_dual_step(okkt_kkt_handle k, const double* J_nzval_cand, const double* grad_cand, const double* s_cand,
                       const double* y_cand, double mu_cand, double a_norm_penalty, double step_size_P, double lb, double ub,
                       int dual_ls, double scale_D, double scale_mu, double* step_size_D) {
  int rc = ls_ready(k, true);
  if (rc != OKKT_OK) return rc;
  if (!step_size_D) return OKKT_ERR_INVALID;
  const double small_step = std::max(lb, std::min(ub, step_size_P));
  if (dual_ls != 1 && dual_ls != 3) { *step_size_D = ub; return OKKT_OK; }
  const int64_t n = k->n, m = k->m;
  if ((n > 0 && !grad_cand) || (m > 0 && (!s_cand || !y_cand))) return OKKT_ERR_INVALID;
  hipStream_t st = kk_stream(k);
  const double* Jx = k->Jx;
  if (J_nzval_cand && k->nnzJ) {
    if (!k->Jcur) {
      void* p = nullptr;
      KK_TRY(k, hipMalloc(&p, (size_t)k->nnzJ * 8));
      k->allocs.push_back(p);
      k->Jcur = (double*)p;
    }
    KK_TRY(k, hipMemcpyAsync(k->Jcur, J_nzval_cand, (size_t)k->nnzJ * 8, hipMemcpyHostToDevice, st));
    Jx = k->Jcur;
  }
  if ((rc = stage(k, k->ls_n[0], grad_cand, n)) != OKKT_OK || (rc = stage(k, k->ls_m[1], s_cand, m)) != OKKT_OK ||
      (rc = stage(k, k->ls_m[2], y_cand, m)) != OKKT_OK)
    return rc;
  double rn[2] = {0.0, 0.0}, rm[2] = {0.0, 0.0};
  Channels<2> ch{{kSum, kSum}, {0.0, 0.0}};
  if (n) {
    if (m) {
      // eval_grad_lag(new_it, mu) = grad - J'y + mu * pen * J'1 = grad - J'(y - mu * pen)
      hipLaunchKernelGGL(k_axpb, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, 1.0, k->ls_m[2], -(mu_cand * a_norm_penalty), k->ls_m[3], 0);
      kk_spmv_JT(k, Jx, k->ls_m[3], k->ls_n[1]);
      kk_spmv_JT(k, Jx, k->dy, k->vn1);
    } else {
      KK_TRY(k, hipMemsetAsync(k->ls_n[1], 0, (size_t)n * 8, st));
      KK_TRY(k, hipMemsetAsync(k->vn1, 0, (size_t)n * 8, st));
    }
    if ((rc = map_reduce<2>(k, n, MapDualStepN{k->ls_n[0], k->ls_n[1], k->vn1, scale_D}, ch, rn)) != OKKT_OK) return rc;
  }
  if (m && (rc = map_reduce<2>(k, m, MapDualStepM{k->ls_m[1], k->ls_m[2], k->dy, scale_mu, mu_cand}, ch, rm)) != OKKT_OK) return rc;
  double sd = (rn[0] + rm[0]) / (rn[1] + rm[1]);
  sd = jl_min(sd, ub);
  sd = jl_max(sd, small_step);
  *step_size_D = sd;
  return OKKT_OK;
}

}  // extern "C"
