// C ABI, KKT-system level (include/okkt.h): device-resident counterpart of the reference's
// Schur_KKT_solver / Symmetric_KKT_solver.
//
//   form_system!                       /root/reference/src/kkt_system_solver/schur.jl:47-62, symmetric.jl:35-53
//   eval_J_T_J / eval_diag_J_T_J       src/utils/eval.jl:85-100          -> k_assemble_schur / k_schur_diag
//   update_delta_vecs! + factor!       schur.jl:64-87, symmetric.jl:55-57,85-102, kkt_system_solver.jl:98-113,190-204
//   System_rhs                         src/kkt_system_solver/system_rhs.jl:57-73
//   compute_direction_implementation!  schur.jl:89-182, symmetric.jl:59-83
//   update_kkt_error!                  kkt_system_solver.jl:27-47,67-96
//   ipopt_strategy!                    src/IPM/delta_strategy.jl:37-114
//   eval_jac_prod / eval_jac_T_prod / hess_product   eval.jl:102-108,221-234 -> k_spmv_* kernels
//
// All sums are owner-computes (one thread per output entry, fixed order): results are bitwise
// reproducible from run to run; there are no floating-point atomics.
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "kkt_state.h"

using namespace okkt;

namespace {

template <typename T>
int kk_upload(okkt_kkt_s* k, const std::vector<T>& v, T** out) {
  void* p = nullptr;
  KK_TRY(k, hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  k->allocs.push_back(p);
  if (!v.empty()) KK_TRY(k, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return OKKT_OK;
}
template <typename T>
int kk_alloc(okkt_kkt_s* k, size_t count, T** out) {
  void* p = nullptr;
  KK_TRY(k, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
  k->allocs.push_back(p);
  KK_TRY(k, hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)));
  KK_TRY(k, hipStreamSynchronize(nullptr));   // the fill runs on the null stream; the handle's streams do not wait for it
  *out = (T*)p;
  return OKKT_OK;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + 255) / 256)); }

// ---- assembly ---------------------------------------------------------------------------------
// K = [[H, J'],[J, -diag(s/y)]], lower triangle only: pure gather/scatter through precomputed slots
__global__ void k_assemble_aug(int64_t nnzH, int64_t nnzJ, int64_t n, int64_t m, const double* __restrict__ Hx,
                               const double* __restrict__ Jx, const double* __restrict__ s, const double* __restrict__ y,
                               const int64_t* __restrict__ mapH, const int64_t* __restrict__ mapJ,
                               const int64_t* __restrict__ diagA, double* __restrict__ A) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nnzH) A[mapH[t]] = Hx[t];
  else if (t < nnzH + nnzJ) A[mapJ[t - nnzH]] = Jx[t - nnzH];
  else if (t < nnzH + nnzJ + m) { const int64_t i = t - nnzH - nnzJ; A[diagA[n + i]] = -s[i] / y[i]; }
}

// Q = J' diag(sig) J + H on the lower triangle: every entry sums its own contribution list
// (row i ascending, like Gustavson's product in eval.jl:85-87), then adds H
__global__ void k_assemble_schur(int64_t nnzQ, const int64_t* __restrict__ qptr, const int* __restrict__ qa,
                                 const int* __restrict__ qb, const int* __restrict__ qi, const int64_t* __restrict__ qh,
                                 const double* __restrict__ Jx, const double* __restrict__ sig,
                                 const double* __restrict__ Hx, double* __restrict__ A) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnzQ) return;
  double v = 0.0;
  for (int64_t t = qptr[e]; t < qptr[e + 1]; ++t) v += (Jx[qa[t]] * sig[qi[t]]) * Jx[qb[t]];
  if (qh[e] >= 0) v += Hx[qh[e]];
  A[e] = v;
}

// The same sum with the column of Q resident in LDS (eval_J_T_J, eval.jl:85-87; form_system!, schur.jl:55).  A group of 16
// lanes owns column b of Q: for every row i of J that holds column b (ascending: the reference's summation order) the lanes
// stream the CSR segment of row i with columns a >= b -- values contiguous in the CSR-ordered copy -- and add
// (J_ia sig_i) J_ib into the column's accumulator at a precomputed 16-bit slot; the column leaves LDS once, with H added.
// 12 bytes per term (8 value + 2 slot + the per-row scalars) instead of 12 bytes of index lists plus three gathers, and
// the read-modify-write chain of an entry runs in LDS.  Bitwise the same result as k_assemble_schur.
__global__ __launch_bounds__(256) void k_assemble_schur_lds(int64_t n, int G, int maxcol, const int64_t* __restrict__ Ap, const int64_t* __restrict__ Jp,
                                                            const int* __restrict__ Ji, const double* __restrict__ Jx, const int64_t* __restrict__ Jrp,
                                                            const double* __restrict__ Jcsr, const int64_t* __restrict__ seg_q,
                                                            const int64_t* __restrict__ seg_t, const uint16_t* __restrict__ tslot,
                                                            const double* __restrict__ sig, const int64_t* __restrict__ qh,
                                                            const double* __restrict__ Hx, double* __restrict__ A) {
  extern __shared__ __attribute__((aligned(16))) double sacc[];
  const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int64_t b = (int64_t)blockIdx.x * G + g;
  if (g >= G || b >= n) return;                  // no workgroup barrier below: a group only talks to itself (one wave)
  double* acc = sacc + (size_t)g * maxcol;
  const int64_t a0 = Ap[b];
  const int ncol = (int)(Ap[b + 1] - a0);
  for (int e = l; e < ncol; e += 16) acc[e] = 0.0;
  __threadfence_block();
  for (int64_t p = Jp[b]; p < Jp[b + 1]; ++p) {
    const int i = Ji[p];
    const double c = sig[i], jb = Jx[p];
    const int64_t q0 = seg_q[p], q1 = Jrp[i + 1], t0 = seg_t[p];
    for (int64_t q = q0 + l; q < q1; q += 16) acc[tslot[t0 + (q - q0)]] += (Jcsr[q] * c) * jb;
    __threadfence_block();                       // the next row may touch the same slots from other lanes of the group
  }
  for (int e = l; e < ncol; e += 16) {
    double v = acc[e];
    const int64_t h = qh[a0 + e];
    if (h >= 0) v += Hx[h];
    A[a0 + e] = v;
  }
}

// schur_diag = diag(H) + sum_i J_ij^2 sig_i (kkt_system_solver.jl:296-300, eval.jl:89-100)
__global__ void k_schur_diag(int64_t n, const int64_t* __restrict__ Jp, const int* __restrict__ Ji,
                             const double* __restrict__ Jx, const double* __restrict__ sig,
                             const int64_t* __restrict__ Hp, const int* __restrict__ Hi,
                             const double* __restrict__ Hx, double* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double di = 0.0;
  for (int64_t p = Jp[j]; p < Jp[j + 1]; ++p) di += Jx[p] * Jx[p] * sig[Ji[p]];
  double h = 0.0;
  for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) if (Hi[p] == j) h += Hx[p];
  out[j] = h + di;
}
__global__ void k_gather(int64_t n, const int64_t* __restrict__ idx, const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_scale(int64_t n, double a, const double* __restrict__ x, double* __restrict__ o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a * x[i];
}
__global__ void k_div(int64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] / b[i];
}

// ---- SpMV -------------------------------------------------------------------------------------
// y = L x + L' x - diag(L) x for a lower-triangular CSC L (hess_product, eval.jl:221-234)
__global__ void k_spmv_symlower(int64_t n, const int64_t* __restrict__ cp, const int* __restrict__ ri,
                                const double* __restrict__ vals, const int64_t* __restrict__ rp,
                                const int* __restrict__ rj, const int64_t* __restrict__ rmap,
                                const double* __restrict__ x, double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v1 = 0.0, v2 = 0.0, dg = 0.0;
  for (int64_t p = rp[i]; p < rp[i + 1]; ++p) {          // row i of L
    v1 += vals[rmap[p]] * x[rj[p]];
    if (rj[p] == i) dg += vals[rmap[p]];
  }
  for (int64_t p = cp[i]; p < cp[i + 1]; ++p) v2 += vals[p] * x[ri[p]];   // column i of L = row i of L'
  y[i] = v1 + v2 - dg * x[i];
}

// ---- vector kernels -----------------------------------------------------------------------------
__global__ void k_schur_t1(int64_t m, const double* rP, const double* rC, const double* sig, const double* s, double* o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) o[i] = rP[i] * sig[i] + rC[i] / s[i];                 // schur.jl:103
}
__global__ void k_sym_rhs(int64_t n, int64_t m, const double* rD, const double* rP, const double* rC, const double* y, double* o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = rD[i];
  else if (i < n + m) o[i] = rP[i - n] + rC[i - n] / y[i - n];       // symmetric.jl:65
}
__global__ void k_sym_split(int64_t n, int64_t m, const double* sol, double* dx, double* dy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = sol[i];
  else if (i < n + m) dy[i - n] = -sol[i];                           // symmetric.jl:72-73
}
// system_rhs.jl:57-73 + eval.jl:59-63,136-142
__global__ void k_rhs_pc(int64_t m, const double* cons, const double* s, const double* y, double one_minus_P, double mu_target, double* rP, double* rC) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) { rP[i] = -(cons[i] - s[i]) * one_minus_P; rC[i] = mu_target - s[i] * y[i]; }
}

// single-block reductions (inputs are O(n+m) vectors; deterministic)
__global__ __launch_bounds__(1024) void k_reduce(int64_t n, const double* __restrict__ v, int mode /*0 min, 1 max|.|*/, double* out) {
  __shared__ double sh[1024];
  double a = mode == 0 ? INFINITY : 0.0;
  bool nan = false;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double x = v[i];
    if (x != x) nan = true;
    if (mode == 0) a = fmin(a, x); else a = fmax(a, fabs(x));
  }
  sh[threadIdx.x] = nan ? NAN : a;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const double p = sh[threadIdx.x], q = sh[threadIdx.x + o];
      sh[threadIdx.x] = (p != p || q != q) ? NAN : (mode == 0 ? fmin(p, q) : fmax(p, q));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = n > 0 ? sh[0] : (mode == 0 ? INFINITY : 0.0);
}


// ---- clever symmetric ------------------------------------------------------------------------------------
// update_indicies! (clever_symmetric.jl:262-287): one thread per group, members in ls order
__global__ void k_clever_groups(int64_t m_new, const int64_t* __restrict__ gptr, const int* __restrict__ mind,
                                const double* __restrict__ mratio, const double* __restrict__ s, const double* __restrict__ y,
                                double* __restrict__ gU, double* __restrict__ rowg) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= m_new) return;
  double u_inv = 0.0;
  for (int64_t t = gptr[g]; t < gptr[g + 1]; ++t) { const int j = mind[t]; const double u = s[j] / y[j]; u_inv += (mratio[t] * mratio[t]) * (1.0 / u); }
  const double U = 1.0 / u_inv;
  gU[g] = U;
  for (int64_t t = gptr[g]; t < gptr[g + 1]; ++t) { const int j = mind[t]; const double u = s[j] / y[j]; rowg[j] = U * mratio[t] * (1.0 / u); }
}
// diag_rescale (clever_symmetric.jl:307-319)
__global__ void k_clever_rescale(int64_t n, int64_t m_new, int mode, double mu, double xscale, const double* __restrict__ gU, double* __restrict__ D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + m_new) return;
  if (i < n) D[i] = mode == OKKT_RESCALE_U_AND_X ? xscale : 1.0;
  else D[i] = mode == OKKT_RESCALE_NONE ? 1.0 : mu / sqrt(gU[i - n]);
}
// Q = D M D, M = [[H 0];[J_new -U_new]] (lower): gather through precomputed slots, then the (2,2) diagonal
__global__ void k_clever_assemble(int64_t nnzH, int64_t nnzJ, int64_t n, int64_t m_new, const double* __restrict__ Hx,
                                  const double* __restrict__ Jx, const int64_t* __restrict__ mapH, const int64_t* __restrict__ mapJc,
                                  const int* __restrict__ Hcol, const int* __restrict__ Jcol,
                                  const int* __restrict__ Ai, const int64_t* __restrict__ diagA, const double* __restrict__ gU,
                                  const double* __restrict__ D, double* __restrict__ A) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nnzH) { const int64_t e = mapH[t]; A[e] = (D[Ai[e]] * Hx[t]) * D[Hcol[t]]; }
  else if (t < nnzH + nnzJ) {
    const int64_t p = t - nnzH, e = mapJc[p];
    if (e >= 0) A[e] = (D[Ai[e]] * Jx[p]) * D[Jcol[p]];
  } else if (t < nnzH + nnzJ + m_new) {
    const int64_t g = t - nnzH - nnzJ;
    A[diagA[n + g]] = (D[n + g] * -gU[g]) * D[n + g];
  }
}
// true_x_diag = diag(M)[1:n] (unscaled H diagonal)
__global__ void k_clever_true_diag(int64_t n, const int64_t* __restrict__ Hp, const int* __restrict__ Hi, const double* __restrict__ Hx, double* __restrict__ o) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double v = 0.0;
  for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) if (Hi[p] == j) v += Hx[p];
  o[j] = v;
}
// update_delta_vecs! (clever_symmetric.jl:494-519): with sum(abs.(delta_x_vec)) > 0, i.e. delta != 0 (either sign), the x
// diagonal becomes the UNscaled true_x_diag (+ delta, added by the factorisation's shift), with delta == 0 it stays
// D^2 * true_x_diag
__global__ void k_clever_xdiag(int64_t n, int delta_pos, const double* __restrict__ D, const double* __restrict__ tx, const int64_t* __restrict__ diagA, double* __restrict__ A) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) A[diagA[j]] = delta_pos ? tx[j] : (D[j] * tx[j]) * D[j];
}
__global__ void k_clever_symrhs(int64_t m, const double* rP, const double* rC, const double* y, double* o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) o[i] = rP[i] + rC[i] / y[i];
}
__global__ void k_clever_crhs(int64_t m_new, const int64_t* __restrict__ gptr, const int* __restrict__ mind, const double* __restrict__ rowg,
                              const double* __restrict__ symrhs, double* __restrict__ crhs) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= m_new) return;
  double v = 0.0;
  for (int64_t t = gptr[g]; t < gptr[g + 1]; ++t) { const int j = mind[t]; v += rowg[j] * symrhs[j]; }
  crhs[g] = v;
}
__global__ void k_clever_rhs(int64_t n, int64_t m_new, const double* rD, const double* crhs, const double* D, double* o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = rD[i] * D[i];
  else if (i < n + m_new) o[i] = crhs[i - n] * D[i];
}
// err = rhs - (Q sol), Q carrying delta on its first n diagonal entries
__global__ void k_clever_res(int64_t dim, int64_t n, double delta, const double* rhs, const double* qx, const double* sol, double* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < dim) err[i] = rhs[i] - (qx[i] + (i < n ? delta * sol[i] : 0.0));
}
__global__ void k_clever_unscale(int64_t n, int64_t m_new, const double* sol, const double* D, double* dx, double* v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = sol[i] * D[i];
  else if (i < n + m_new) v[i - n] = sol[i] * D[i];
}
// dir.y (clever_symmetric.jl:457-474): y_j = u_j^-1 symrhs_j + u_j^-1 ratio_j * (-(crhs_g + U_g v_g)), g = group of row j
__global__ void k_clever_y(int64_t m, const int* __restrict__ row_grp, const double* __restrict__ row_ratio, const double* __restrict__ s,
                           const double* __restrict__ y, const double* __restrict__ symrhs, const double* __restrict__ crhs,
                           const double* __restrict__ gU, const double* __restrict__ v, double* __restrict__ dy) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const double uinv = 1.0 / (s[j] / y[j]);
  const int g = row_grp[j];
  const double tmp = -(crhs[g] + gU[g] * v[g]);
  dy[j] = uinv * symrhs[j] + (uinv * row_ratio[j]) * tmp;
}

// ---- segmented sparse products -----------------------------------------------------------------------------------
// A row (CSR) or column (CSC) of J / H holds 10 - 40 entries at the BASELINE sizes: one thread per row reads its values
// with a stride of a whole row between neighbouring lanes (uncoalesced), one wave per row idles most lanes.  Here a group
// of LPR lanes (4 ... 64, chosen from the average length) owns a row: consecutive groups of a wave own consecutive rows,
// so a wave reads one contiguous run of values and indices; the partial sums meet by xor-shuffles inside the group.
// The summation order is fixed by LPR alone (bitwise reproducible run to run).  No early return before the shuffles:
// rows past the end are clamped and simply not stored.
template <int LPR>
__device__ __forceinline__ double seg_dot(const int64_t* __restrict__ ptr, const int* __restrict__ idx, const double* __restrict__ vals,
                                          const double* __restrict__ x, int64_t row, int sub) {
  double a = 0.0;
  const int64_t p1 = ptr[row + 1];
  for (int64_t p = ptr[row] + sub; p < p1; p += LPR) a += vals[p] * x[idx[p]];
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  return a;
}
// hess_product (eval.jl:221-234) of row i: (L x)_i + (L' x)_i - diag(L)_i x_i with the lower-stored H in CSR and CSC order
template <int LPR>
__device__ __forceinline__ double seg_hess(const int64_t* __restrict__ Hrp, const int* __restrict__ Hrj, const double* __restrict__ Hcsr,
                                           const int64_t* __restrict__ Hp, const int* __restrict__ Hi, const double* __restrict__ Hx,
                                           const double* __restrict__ Hdiag, const double* __restrict__ x, int64_t i, int sub) {
  const double v1 = seg_dot<LPR>(Hrp, Hrj, Hcsr, x, i, sub);
  const double v2 = seg_dot<LPR>(Hp, Hi, Hx, x, i, sub);
  return (v1 + v2) - Hdiag[i] * x[i];
}
__device__ __forceinline__ double nan_max(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }
// per-workgroup maxima of NV values per thread (NaN propagates like Julia's norm(., Inf)); slot v of workgroup b -> part[b * 8 + v]
template <int NV>
__device__ __forceinline__ void block_max_store(double (&v)[NV], double* __restrict__ part) {
  __shared__ double sh[NV][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double a = v[q];
    for (int o = 32; o > 0; o >>= 1) a = nan_max(a, __shfl_xor(a, o, 64));
    if (lane == 0) sh[q][wv] = a;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int q = threadIdx.x;
    part[(size_t)blockIdx.x * 8 + q] = nan_max(nan_max(sh[q][0], sh[q][1]), nan_max(sh[q][2], sh[q][3]));
  }
}

// y[i] = dot_i (* scale[i]) (+ beta * addv[i]): J x, J' v, Sigma .* (J x), rD + J' t, J dx - rP
// is_diag_dom (delta_strategy.jl:1-9): margin[i] = 3 Q[i,i] - (sum(Q[:,i]) + sum(Q[i,:])) over the STORED entries of the x-block
// (H lower-stored; the Schur matrix carries the full J' Sigma J on top).  hcol / hrow: column / row sums of the stored H,
// jsj: row sum (= column sum) of J' Sigma J or NULL, qdiag: the block's diagonal without delta.  NaN compares false in Julia.
__global__ void k_diag_dom_margin(int64_t n, const double* __restrict__ hcol, const double* __restrict__ hrow, const double* __restrict__ jsj,
                                  const double* __restrict__ qdiag, double delta, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = qdiag[i] + delta;
  const double cs = hcol[i] + (jsj ? jsj[i] : 0.0) + delta, rs = hrow[i] + (jsj ? jsj[i] : 0.0) + delta;
  const double v = 3.0 * d - (cs + rs);
  out[i] = (v != v) ? 1.0e308 : v;
}

template <int LPR>
__global__ __launch_bounds__(256) void k_seg_spmv(int64_t nrows, const int64_t* __restrict__ ptr, const int* __restrict__ idx,
                                                  const double* __restrict__ vals, const double* __restrict__ x, const double* __restrict__ scale,
                                                  const double* __restrict__ addv, double beta, double* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t rc = row < nrows ? row : nrows - 1;
  double r = seg_dot<LPR>(ptr, idx, vals, x, rc, sub);
  if (sub == 0 && row < nrows) {
    if (scale) r *= scale[row];
    if (addv) r = r + beta * addv[row];
    y[row] = r;
  }
}
template <int LPR>
__global__ __launch_bounds__(256) void k_seg_hess(int64_t n, const int64_t* __restrict__ Hrp, const int* __restrict__ Hrj, const double* __restrict__ Hcsr,
                                                  const int64_t* __restrict__ Hp, const int* __restrict__ Hi, const double* __restrict__ Hx,
                                                  const double* __restrict__ Hdiag, const double* __restrict__ x, double* __restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t i = row < n ? row : n - 1;
  const double h = seg_hess<LPR>(Hrp, Hrj, Hcsr, Hp, Hi, Hx, Hdiag, x, i, sub);
  if (sub == 0 && row < n) y[i] = h;
}
// res = rhs - (J' v + (H dx + delta dx)) with v = Sigma .* (J dx) (schur.jl:166-170): one pass instead of J' v, H dx and
// the vector kernel
template <int LPR>
__global__ __launch_bounds__(256) void k_schur_resid(int64_t n, const int64_t* __restrict__ Jp, const int* __restrict__ Ji, const double* __restrict__ Jx,
                                                     const double* __restrict__ v, const int64_t* __restrict__ Hrp, const int* __restrict__ Hrj,
                                                     const double* __restrict__ Hcsr, const int64_t* __restrict__ Hp, const int* __restrict__ Hi,
                                                     const double* __restrict__ Hx, const double* __restrict__ Hdiag, const double* __restrict__ dx,
                                                     const double* __restrict__ rhs, double delta, double* __restrict__ res) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t i = row < n ? row : n - 1;
  const double jac = seg_dot<LPR>(Jp, Ji, Jx, v, i, sub);
  const double hx = seg_hess<LPR>(Hrp, Hrj, Hcsr, Hp, Hi, Hx, Hdiag, dx, i, sub);
  if (sub == 0 && row < n) res[i] = rhs[i] - (jac + (hx + delta * dx[i]));
}
// dy, ds from J dx (schur.jl:113-116); direct = 1: ds = (comp_r - dy .* s) ./ y (schur_direct.jl:54-56), y, s, sig of the CURRENT iterate
template <int LPR>
__global__ __launch_bounds__(256) void k_schur_dyds(int64_t m, const int64_t* __restrict__ Jrp, const int* __restrict__ Jrj, const double* __restrict__ Jcsr,
                                                    const double* __restrict__ dx, const double* __restrict__ rP, const double* __restrict__ rC,
                                                    const double* __restrict__ y, const double* __restrict__ s, const double* __restrict__ sig, int direct,
                                                    double* __restrict__ dy, double* __restrict__ ds) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t i = row < m ? row : m - 1;
  const double Jdx = seg_dot<LPR>(Jrp, Jrj, Jcsr, dx, i, sub);
  if (sub == 0 && row < m) {
    const double dyi = -(Jdx - (rP[i] + rC[i] / y[i])) * sig[i];
    dy[i] = dyi;
    ds[i] = direct ? (rC[i] - dyi * s[i]) / y[i] : Jdx - rP[i];
  }
}
// update_kkt_error! (kkt_system_solver.jl:27-47,67-96), dual block: |predicted_lag_change - dual_r| and |dual_r|, maxima per workgroup
template <int LPR>
__global__ __launch_bounds__(256) void k_err_dual(int64_t n, const int64_t* __restrict__ Jp, const int* __restrict__ Ji, const double* __restrict__ Jx,
                                                  const double* __restrict__ dy, const int64_t* __restrict__ Hrp, const int* __restrict__ Hrj,
                                                  const double* __restrict__ Hcsr, const int64_t* __restrict__ Hp, const int* __restrict__ Hi,
                                                  const double* __restrict__ Hx, const double* __restrict__ Hdiag, const double* __restrict__ dx,
                                                  const double* __restrict__ rD, double delta, double* __restrict__ part) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t i = row < n ? row : n - 1;
  const double jty = seg_dot<LPR>(Jp, Ji, Jx, dy, i, sub);
  const double hx = seg_hess<LPR>(Hrp, Hrj, Hcsr, Hp, Hi, Hx, Hdiag, dx, i, sub);
  double v[2] = {0.0, 0.0};
  if (sub == 0 && row < n) {
    v[0] = fabs(((delta * dx[i] + 0.0) + hx - jty) - rD[i]);
    v[1] = fabs(rD[i]);
  }
  block_max_store<2>(v, part);
}
// primal and complementarity blocks: |J dx - ds - primal_r|, |s dy + y ds - comp_r|, |primal_r|, |comp_r|
template <int LPR>
__global__ __launch_bounds__(256) void k_err_pc(int64_t m, const int64_t* __restrict__ Jrp, const int* __restrict__ Jrj, const double* __restrict__ Jcsr,
                                                const double* __restrict__ dx, const double* __restrict__ ds, const double* __restrict__ dy,
                                                const double* __restrict__ s, const double* __restrict__ y, const double* __restrict__ rP,
                                                const double* __restrict__ rC, double* __restrict__ part) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t i = row < m ? row : m - 1;
  const double Jdx = seg_dot<LPR>(Jrp, Jrj, Jcsr, dx, i, sub);
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (sub == 0 && row < m) {
    v[0] = fabs(Jdx - ds[i] - rP[i]);
    v[1] = fabs(s[i] * dy[i] + y[i] * ds[i] - rC[i]);
    v[2] = fabs(rP[i]);
    v[3] = fabs(rC[i]);
  }
  block_max_store<4>(v, part);
}
// the six norms: out[0..1] from the nbD dual partial rows, out[2..5] from the nbM rows behind them
__global__ __launch_bounds__(256) void k_err_final(int64_t nbD, int64_t nbM, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double sh[6][256];
  double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t b = threadIdx.x; b < nbD; b += 256) { a[0] = nan_max(a[0], part[b * 8]); a[1] = nan_max(a[1], part[b * 8 + 1]); }
  for (int64_t b = threadIdx.x; b < nbM; b += 256)
    for (int q = 0; q < 4; ++q) a[2 + q] = nan_max(a[2 + q], part[(nbD + b) * 8 + q]);
  for (int q = 0; q < 6; ++q) sh[q][threadIdx.x] = a[q];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 6; ++q) sh[q][threadIdx.x] = nan_max(sh[q][threadIdx.x], sh[q][threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x < 6) out[threadIdx.x] = sh[threadIdx.x][0];
}
// System_rhs, dual block (system_rhs.jl:57-73 + eval.jl:59-63,136-142): J' y and J' 1 in one pass over the column
template <int LPR>
__global__ __launch_bounds__(256) void k_rhs_dual_seg(int64_t n, const int64_t* __restrict__ Jp, const int* __restrict__ Ji, const double* __restrict__ Jx,
                                                      const double* __restrict__ y, const double* __restrict__ grad, double mu_pen, double one_minus_D,
                                                      double* __restrict__ o) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t i = row < n ? row : n - 1;
  double jty = 0.0, jt1 = 0.0;
  const int64_t p1 = Jp[i + 1];
  for (int64_t p = Jp[i] + sub; p < p1; p += LPR) { const double a = Jx[p]; jty += a * y[Ji[p]]; jt1 += a; }
#pragma unroll
  for (int q = LPR / 2; q > 0; q >>= 1) { jty += __shfl_xor(jty, q, 64); jt1 += __shfl_xor(jt1, q, 64); }
  if (sub == 0 && row < n) o[i] = -((grad[i] - jty) + mu_pen * jt1) * one_minus_D;
}
// schur_diag = diag(H) + sum_i J_ij^2 sig_i (kkt_system_solver.jl:296-300, eval.jl:89-100)
template <int LPR>
__global__ __launch_bounds__(256) void k_schur_diag_seg(int64_t n, const int64_t* __restrict__ Jp, const int* __restrict__ Ji, const double* __restrict__ Jx,
                                                        const double* __restrict__ sig, const double* __restrict__ Hdiag, double* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / LPR;
  const int sub = (int)(t % LPR);
  const int64_t j = row < n ? row : n - 1;
  double di = 0.0;
  const int64_t p1 = Jp[j + 1];
  for (int64_t p = Jp[j] + sub; p < p1; p += LPR) di += Jx[p] * Jx[p] * sig[Ji[p]];
#pragma unroll
  for (int q = LPR / 2; q > 0; q >>= 1) di += __shfl_xor(di, q, 64);
  if (sub == 0 && row < n) out[j] = Hdiag[j] + di;
}
// values of J and H in CSR order, Sigma = y ./ s and diag(H), in one launch behind the uploads of form_system
__global__ void k_form_prep(int64_t nnzJ, int64_t nnzH, int64_t n, int64_t m, const double* __restrict__ Jx, const int64_t* __restrict__ Jrmap,
                            const double* __restrict__ Hx, const int64_t* __restrict__ Hrmap, const int64_t* __restrict__ Hp, const int* __restrict__ Hi,
                            const double* __restrict__ s, const double* __restrict__ y, double* __restrict__ Jcsr, double* __restrict__ Hcsr,
                            double* __restrict__ Hdiag, double* __restrict__ sig) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nnzJ) Jcsr[t] = Jx[Jrmap[t]];
  if (t < nnzH) Hcsr[t] = Hx[Hrmap[t]];
  if (t < m) sig[t] = y[t] / s[t];
  if (t < n) {
    double v = 0.0;
    for (int64_t p = Hp[t]; p < Hp[t + 1]; ++p) if (Hi[p] == t) v += Hx[p];
    Hdiag[t] = v;
  }
}

inline int pick_lpr(int64_t nnz, int64_t rows) {
  const double avg = rows > 0 ? (double)nnz / (double)rows : 1.0;
  return avg <= 5.0 ? 4 : (avg <= 10.0 ? 8 : (avg <= 24.0 ? 16 : (avg <= 56.0 ? 32 : 64)));
}
inline dim3 seg_grid(int64_t rows, int lpr) { return dim3((unsigned)std::max<int64_t>(1, (rows * lpr + 255) / 256)); }
#define SEG_LAUNCH(KERN, lpr, rows, st, ...)                                                                   \
  do {                                                                                                         \
    const int64_t rows__ = (rows);                                                                             \
    if (rows__ > 0) switch (lpr) {                                                                             \
        case 4: hipLaunchKernelGGL((KERN<4>), seg_grid(rows__, 4), dim3(256), 0, st, __VA_ARGS__); break;      \
        case 8: hipLaunchKernelGGL((KERN<8>), seg_grid(rows__, 8), dim3(256), 0, st, __VA_ARGS__); break;      \
        case 16: hipLaunchKernelGGL((KERN<16>), seg_grid(rows__, 16), dim3(256), 0, st, __VA_ARGS__); break;   \
        case 32: hipLaunchKernelGGL((KERN<32>), seg_grid(rows__, 32), dim3(256), 0, st, __VA_ARGS__); break;   \
        default: hipLaunchKernelGGL((KERN<64>), seg_grid(rows__, 64), dim3(256), 0, st, __VA_ARGS__); break;   \
      }                                                                                                        \
  } while (0)

int kk_fail(okkt_kkt_s* k, int code, const std::string& msg) { k->err = msg; return code; }

int kk_check_ls(okkt_kkt_s* k, int rc, const char* what) {
  if (rc < 0) { k->err = std::string(what) + ": " + okkt_last_error(k->ls); }
  return rc;
}


int kk_reduce(okkt_kkt_s* k, int64_t n, const double* v, int mode, double* host_out) {
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, kk_stream(k), n, v, mode, k->red);
  KK_TRY(k, hipMemcpyAsync(host_out, k->red, sizeof(double), hipMemcpyDeviceToHost, kk_stream(k)));
  KK_TRY(k, hipStreamSynchronize(kk_stream(k)));
  return OKKT_OK;
}

// J x, J' v, H x on the handle's stream (segmented products; J x and H x read the CSR-ordered copies of form_system)
void spmv_J(okkt_kkt_s* k, const double* x, double* y) {
  SEG_LAUNCH(k_seg_spmv, k->lprJr, k->m, kk_stream(k), k->m, k->Jrp, k->Jrj, k->Jcsr, x, (const double*)nullptr, (const double*)nullptr, 0.0, y);
}
void spmv_JT(okkt_kkt_s* k, const double* Jx, const double* v, double* y) {
  SEG_LAUNCH(k_seg_spmv, k->lprJc, k->n, kk_stream(k), k->n, k->Jp, k->Ji, Jx, v, (const double*)nullptr, (const double*)nullptr, 0.0, y);
}
void spmv_H(okkt_kkt_s* k, const double* x, double* y) {
  SEG_LAUNCH(k_seg_hess, k->lprH, k->n, kk_stream(k), k->n, k->Hrp, k->Hrj, k->Hcsr, k->Hp, k->Hi, k->Hx, k->Hdiag, x, y);
}

}  // namespace

namespace okkt {
void kk_spmv_J(okkt_kkt_s* k, const double* x, double* y) { spmv_J(k, x, y); }
void kk_spmv_JT(okkt_kkt_s* k, const double* Jx, const double* v, double* y) { spmv_JT(k, Jx, v, y); }
void kk_spmv_H(okkt_kkt_s* k, const double* x, double* y) { spmv_H(k, x, y); }
}  // namespace okkt

extern "C" {

int okkt_kkt_default_pars(okkt_kkt_pars* p) {
  if (!p) return OKKT_ERR_INVALID;
  p->delta_start = 1e-6;                 // parameters.jl:147-158
  p->delta_min = 1e-12;
  p->delta_max = 1e50;
  p->delta_inc = 8.0;
  p->delta_dec = 1.0 / M_PI;
  p->delta_zero = 0.0;
  p->ItRefine_Num = 3;                   // parameters.jl:20
  p->max_it = 500;                       // delta_strategy.jl:40
  return OKKT_OK;
}

int okkt_kkt_create(okkt_kkt_handle* out, const okkt_opts* opts, int kkt_kind) {
  if (!out) return OKKT_ERR_INVALID;
  *out = nullptr;
  if (kkt_kind != OKKT_KKT_SCHUR && kkt_kind != OKKT_KKT_SYMMETRIC && kkt_kind != OKKT_KKT_CLEVER_SYMMETRIC && kkt_kind != OKKT_KKT_SCHUR_DIRECT) return OKKT_ERR_INVALID;
  okkt_kkt_s* k = new (std::nothrow) okkt_kkt_s();
  if (!k) return OKKT_ERR_ALLOC;
  k->kind = kkt_kind;
  int rc = okkt_create(&k->ls, opts);
  if (rc != OKKT_OK) { delete k; return rc; }
  *out = k;
  return OKKT_OK;
}

int okkt_kkt_destroy(okkt_kkt_handle k) {
  if (!k) return OKKT_ERR_INVALID;
  if (k->ls && k->ls->device_ready) { (void)hipSetDevice(k->ls->device); (void)hipStreamSynchronize(k->ls->stream); }
  for (void* p : k->allocs) (void)hipFree(p);
  k->tm_form.destroy(); k->tm_factor.destroy(); k->tm_rhs.destroy(); k->tm_dir.destroy();
  if (k->ls) okkt_destroy(k->ls);
  delete k;
  return OKKT_OK;
}

const char* okkt_kkt_last_error(okkt_kkt_handle k) { return k ? k->err.c_str() : "null handle"; }
okkt_handle okkt_kkt_linear_solver(okkt_kkt_handle k) { return k ? k->ls : nullptr; }

int okkt_kkt_set_structure(okkt_kkt_handle k, int64_t n, int64_t m, const int64_t* H_colptr, const int64_t* H_rowval,
                           const int64_t* J_colptr, const int64_t* J_rowval, int index_base) {
  if (!k || !H_colptr || !J_colptr || n < 0 || m < 0) return OKKT_ERR_INVALID;
  if (index_base != 0 && index_base != 1) return kk_fail(k, OKKT_ERR_INVALID, "index_base must be 0 or 1");
  if (k->ls->opts.host_symbolic_only || !k->ls->device_ready) return kk_fail(k, OKKT_ERR_NO_DEVICE, "no HIP device");
  if (k->structured) return kk_fail(k, OKKT_ERR_INVALID, "structure already set: create a new handle for a new pattern");
  try {
    const int64_t b = index_base;
    const int64_t nnzH = H_colptr[n] - b, nnzJ = J_colptr[n] - b;
    if (nnzH > 0 && !H_rowval) return OKKT_ERR_INVALID;
    if (nnzJ > 0 && !J_rowval) return OKKT_ERR_INVALID;
    k->n = n; k->m = m; k->nnzH = nnzH; k->nnzJ = nnzJ;
    std::vector<int64_t> Hp(n + 1), Jp(n + 1);
    std::vector<int> Hi(nnzH), Ji(nnzJ);
    for (int64_t j = 0; j <= n; ++j) { Hp[j] = H_colptr[j] - b; Jp[j] = J_colptr[j] - b; }
    for (int64_t j = 0; j < n; ++j) {
      for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) {
        const int64_t i = H_rowval[p] - b;
        if (i < j || i >= n) return kk_fail(k, OKKT_ERR_INVALID, "H must hold the lower triangle only (Class_cutest.jl:548)");
        Hi[p] = (int)i;
      }
      for (int64_t p = Jp[j]; p < Jp[j + 1]; ++p) {
        const int64_t i = J_rowval[p] - b;
        if (i < 0 || i >= m) return kk_fail(k, OKKT_ERR_INVALID, "J row index out of range");
        Ji[p] = (int)i;
      }
    }
    // CSR views
    auto csr_view = [](int64_t nrows, int64_t ncols, const std::vector<int64_t>& cp, const std::vector<int>& ri,
                       std::vector<int64_t>& rp, std::vector<int>& rj, std::vector<int64_t>& rmap) {
      rp.assign(nrows + 1, 0);
      for (size_t p = 0; p < ri.size(); ++p) ++rp[ri[p] + 1];
      for (int64_t i = 0; i < nrows; ++i) rp[i + 1] += rp[i];
      rj.resize(ri.size()); rmap.resize(ri.size());
      std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
      for (int64_t j = 0; j < ncols; ++j)
        for (int64_t p = cp[j]; p < cp[j + 1]; ++p) { const int64_t q = fill[ri[p]]++; rj[q] = (int)j; rmap[q] = p; }
    };
    std::vector<int64_t> Hrp, Hrmap, Jrp, Jrmap;
    std::vector<int> Hrj, Jrj;
    csr_view(n, n, Hp, Hi, Hrp, Hrj, Hrmap);
    csr_view(m, n, Jp, Ji, Jrp, Jrj, Jrmap);
    int rc;
#define UPL(dst, vec) if ((rc = kk_upload(k, vec, &k->dst)) != OKKT_OK) return rc
    UPL(Hp, Hp); UPL(Hi, Hi); UPL(Hrp, Hrp); UPL(Hrj, Hrj); UPL(Hrmap, Hrmap);
    UPL(Jp, Jp); UPL(Ji, Ji); UPL(Jrp, Jrp); UPL(Jrj, Jrj); UPL(Jrmap, Jrmap);
    // ---- pattern of the matrix that is factorised
    if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC) {
      // the reduced pattern depends on the values of J (parallel rows): built by okkt_kkt_compute_indicies
      k->h_Hp = Hp; k->h_Hi = Hi; k->h_Jp = Jp; k->h_Ji = Ji; k->h_Jrp = Jrp; k->h_Jrj = Jrj; k->h_Jrmap = Jrmap;
      std::vector<int> Hcol(nnzH), Jcol(nnzJ);
      for (int64_t j = 0; j < n; ++j) {
        for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) Hcol[p] = (int)j;
        for (int64_t p = Jp[j]; p < Jp[j + 1]; ++p) Jcol[p] = (int)j;
      }
      UPL(Hcol, Hcol); UPL(Jcol, Jcol);
      k->dimA = 0; k->nnzA = 0;
    } else if (k->kind == OKKT_KKT_SYMMETRIC) {
      const int64_t dim = n + m;
      k->dimA = dim;
      std::vector<int64_t> Ap(dim + 1, 0), Ai, mapH(nnzH), mapJ(nnzJ), diagA(dim);
      Ai.reserve(nnzH + nnzJ + dim);
      for (int64_t j = 0; j < n; ++j) {
        // column j: diagonal (always present: delta lands there), H below it, then the J rows
        bool has_diag = false;
        for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) has_diag |= Hi[p] == j;
        diagA[j] = (int64_t)Ai.size();
        if (!has_diag) Ai.push_back(j);
        for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) {
          if (Hi[p] == j) diagA[j] = (int64_t)Ai.size();
          mapH[p] = (int64_t)Ai.size();
          Ai.push_back(Hi[p]);
        }
        for (int64_t p = Jp[j]; p < Jp[j + 1]; ++p) { mapJ[p] = (int64_t)Ai.size(); Ai.push_back(n + Ji[p]); }
        Ap[j + 1] = (int64_t)Ai.size();
      }
      for (int64_t i = 0; i < m; ++i) { diagA[n + i] = (int64_t)Ai.size(); Ai.push_back(n + i); Ap[n + i + 1] = (int64_t)Ai.size(); }
      k->nnzA = (int64_t)Ai.size();
      UPL(mapH, mapH); UPL(mapJ, mapJ); UPL(diagA, diagA);
      k->Ap = Ap; k->Ai = Ai;
    } else {
      // lower triangle of J' S J + H: pairs (a >= b) of the columns present in each row of J
      k->dimA = n;
      std::vector<std::vector<int>> cols(n);   // per column b: rows a >= b
      for (int64_t i = 0; i < m; ++i)
        for (int64_t p = Jrp[i]; p < Jrp[i + 1]; ++p)
          for (int64_t q = Jrp[i]; q < Jrp[i + 1]; ++q)
            if (Jrj[p] >= Jrj[q]) cols[Jrj[q]].push_back(Jrj[p]);
      for (int64_t j = 0; j < n; ++j) {
        cols[j].push_back((int)j);  // diagonal always present (delta)
        for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) cols[j].push_back(Hi[p]);
        std::sort(cols[j].begin(), cols[j].end());
        cols[j].erase(std::unique(cols[j].begin(), cols[j].end()), cols[j].end());
      }
      std::vector<int64_t> Ap(n + 1, 0), Ai, diagA(n);
      for (int64_t j = 0; j < n; ++j) { Ap[j + 1] = Ap[j] + (int64_t)cols[j].size(); }
      Ai.resize(Ap[n]);
      for (int64_t j = 0; j < n; ++j) { std::copy(cols[j].begin(), cols[j].end(), Ai.begin() + Ap[j]); diagA[j] = Ap[j]; }
      k->nnzA = Ap[n];
      std::vector<int64_t> qh(k->nnzA, -1);
      auto slot = [&](int a, int bcol) -> int64_t {
        auto it = std::lower_bound(Ai.begin() + Ap[bcol], Ai.begin() + Ap[bcol + 1], (int64_t)a);
        return it - Ai.begin();
      };
      for (int64_t j = 0; j < n; ++j)
        for (int64_t p = Hp[j]; p < Hp[j + 1]; ++p) qh[slot(Hi[p], (int)j)] = p;
      int64_t maxcol = 1;
      for (int64_t j = 0; j < n; ++j) maxcol = std::max(maxcol, Ap[j + 1] - Ap[j]);
      // LDS-staged assembly (k_assemble_schur_lds) while a column of Q fits a 16-lane group's share of 64 KiB of LDS
      const int G = maxcol <= 512 ? 16 : (maxcol <= 1024 ? 8 : (maxcol <= 2048 ? 4 : 0));
      k->schur_groups = G; k->schur_maxcol = (int)maxcol;
      if (G > 0) {
        // per CSC entry (i, b): where the columns >= b start in row i (rows are visited with growing b: one cursor per row), its
        // first term, and per term the slot of Q(a, b) inside column b (position map of the column's pattern)
        std::vector<int64_t> seg_q(nnzJ), seg_t(nnzJ + 1, 0), cur(Jrp.begin(), Jrp.end() - 1);
        int64_t nterms = 0;
        for (int64_t bcol = 0; bcol < n; ++bcol)
          for (int64_t p = Jp[bcol]; p < Jp[bcol + 1]; ++p) {
            const int i = Ji[p];
            seg_q[p] = cur[i]++;
            seg_t[p] = nterms;
            nterms += Jrp[i + 1] - seg_q[p];
          }
        seg_t[nnzJ] = nterms;
        std::vector<uint16_t> tslot((size_t)nterms);
        std::vector<int> pos(n, -1);
        for (int64_t bcol = 0; bcol < n; ++bcol) {
          for (int64_t e = Ap[bcol]; e < Ap[bcol + 1]; ++e) pos[Ai[e]] = (int)(e - Ap[bcol]);
          for (int64_t p = Jp[bcol]; p < Jp[bcol + 1]; ++p) {
            const int i = Ji[p];
            for (int64_t q = seg_q[p]; q < Jrp[i + 1]; ++q) tslot[(size_t)(seg_t[p] + (q - seg_q[p]))] = (uint16_t)pos[Jrj[q]];
          }
        }
        UPL(seg_q, seg_q); UPL(seg_t, seg_t); UPL(tslot, tslot); UPL(dAp64, Ap);
      } else {
        // contribution lists per Q entry (columns of Q too long for LDS: one thread per entry, k_assemble_schur)
        std::vector<int64_t> qptr(k->nnzA + 1, 0);
        for (int64_t i = 0; i < m; ++i)
          for (int64_t p = Jrp[i]; p < Jrp[i + 1]; ++p)
            for (int64_t q = Jrp[i]; q < Jrp[i + 1]; ++q)
              if (Jrj[p] >= Jrj[q]) ++qptr[slot(Jrj[p], Jrj[q]) + 1];
        for (int64_t e = 0; e < k->nnzA; ++e) qptr[e + 1] += qptr[e];
        std::vector<int> qa(qptr[k->nnzA]), qb(qptr[k->nnzA]), qi(qptr[k->nnzA]);
        std::vector<int64_t> fill(qptr.begin(), qptr.end() - 1);
        for (int64_t i = 0; i < m; ++i)   // rows ascending: fixed summation order
          for (int64_t p = Jrp[i]; p < Jrp[i + 1]; ++p)
            for (int64_t q = Jrp[i]; q < Jrp[i + 1]; ++q)
              if (Jrj[p] >= Jrj[q]) {
                const int64_t t = fill[slot(Jrj[p], Jrj[q])]++;
                // the reference forms (J_T * D) * J: entry (a, b) sums J[i][a] * d_i * J[i][b] over the rows i
                qa[t] = (int)Jrmap[p]; qb[t] = (int)Jrmap[q]; qi[t] = (int)i;
              }
        UPL(qptr, qptr); UPL(qa, qa); UPL(qb, qb); UPL(qi, qi);
      }
      UPL(qh, qh); UPL(diagA, diagA);
      k->Ap = Ap; k->Ai = Ai;
    }
#undef UPL
    if ((rc = kk_alloc(k, (size_t)nnzH, &k->Hx)) || (rc = kk_alloc(k, (size_t)nnzJ, &k->Jx)) || (rc = kk_alloc(k, (size_t)m, &k->s)) ||
        (rc = kk_alloc(k, (size_t)m, &k->y)) || (rc = kk_alloc(k, (size_t)m, &k->sig)) || (rc = kk_alloc(k, (size_t)k->nnzA, &k->Avals)) ||
        (rc = kk_alloc(k, (size_t)n, &k->schur_diag)) || (rc = kk_alloc(k, (size_t)n, &k->rD)) || (rc = kk_alloc(k, (size_t)m, &k->rP)) ||
        (rc = kk_alloc(k, (size_t)m, &k->rC)) || (rc = kk_alloc(k, (size_t)n, &k->dx)) || (rc = kk_alloc(k, (size_t)m, &k->dy)) ||
        (rc = kk_alloc(k, (size_t)m, &k->ds)) || (rc = kk_alloc(k, (size_t)n, &k->vn1)) || (rc = kk_alloc(k, (size_t)n, &k->vn2)) ||
        (rc = kk_alloc(k, (size_t)n, &k->vn3)) || (rc = kk_alloc(k, (size_t)m, &k->vm1)) || (rc = kk_alloc(k, (size_t)m, &k->vm2)) ||
        (rc = kk_alloc(k, (size_t)(n + m), &k->big1)) || (rc = kk_alloc(k, (size_t)(n + m), &k->big2)) || (rc = kk_alloc(k, (size_t)8, &k->red)) ||
        (rc = kk_alloc(k, (size_t)nnzJ, &k->Jcsr)) || (rc = kk_alloc(k, (size_t)nnzH, &k->Hcsr)) || (rc = kk_alloc(k, (size_t)n, &k->Hdiag)) ||
        (rc = kk_alloc(k, (size_t)m, &k->cur_s)) || (rc = kk_alloc(k, (size_t)m, &k->cur_y)) || (rc = kk_alloc(k, (size_t)m, &k->cur_sig)))
      return rc;
    k->lprJr = pick_lpr(nnzJ, m);
    k->lprJc = pick_lpr(nnzJ, n);
    k->lprH = pick_lpr(nnzH, n);
    // partial maxima of the N-err kernels: one row of 8 per workgroup of the widest launch over n and over m
    k->part_blocks = (int64_t)seg_grid(n, 64).x + (int64_t)seg_grid(m, 64).x;
    if ((rc = kk_alloc(k, (size_t)k->part_blocks * 8, &k->part))) return rc;
    if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC) {
      if ((rc = kk_alloc(k, (size_t)m, &k->rowg)) || (rc = kk_alloc(k, (size_t)n, &k->true_x_diag)) ||
          (rc = kk_alloc(k, (size_t)(n + m), &k->big3)) || (rc = kk_alloc(k, (size_t)(n + m), &k->big4)) ||
          (rc = kk_alloc(k, (size_t)(n + m), &k->Dres)) || (rc = kk_alloc(k, (size_t)m, &k->crhs)) || (rc = kk_alloc(k, (size_t)m, &k->gU)))
        return rc;
      k->structured = true;
      return OKKT_OK;
    }
    rc = okkt_analyze(k->ls, k->dimA, k->Ap.data(), k->Ai.data(), 0);
    if (rc != OKKT_OK) return kk_check_ls(k, rc, "okkt_analyze");
    k->structured = true;
    return OKKT_OK;
  } catch (const std::bad_alloc&) {
    return kk_fail(k, OKKT_ERR_ALLOC, "out of host memory in okkt_kkt_set_structure");
  } catch (...) {
    return kk_fail(k, OKKT_ERR_INTERNAL, "unexpected exception in okkt_kkt_set_structure");
  }
}

int okkt_kkt_form_system(okkt_kkt_handle k, const double* H_nzval, const double* J_nzval, const double* s, const double* y) {
  if (!k || !s || !y) return OKKT_ERR_INVALID;
  if (!k->structured) return kk_fail(k, OKKT_ERR_INVALID, "okkt_kkt_set_structure has not been called");
  if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC && !k->indexed) return kk_fail(k, OKKT_ERR_INVALID, "okkt_kkt_compute_indicies has not been called (initialize!, clever_symmetric.jl:53-61)");
  if ((k->nnzH > 0 && !H_nzval) || (k->nnzJ > 0 && !J_nzval)) return OKKT_ERR_INVALID;
  k->have_dir = false;
  k->have_dxnorm = false;
  hipStream_t st = kk_stream(k);
  KK_TRY(k, hipSetDevice(k->ls->device));
  k->tm_form.reset();
  const size_t e0 = k->tm_form.mark(st);
  if (k->nnzH) KK_TRY(k, hipMemcpyAsync(k->Hx, H_nzval, (size_t)k->nnzH * 8, hipMemcpyHostToDevice, st));
  if (k->nnzJ) KK_TRY(k, hipMemcpyAsync(k->Jx, J_nzval, (size_t)k->nnzJ * 8, hipMemcpyHostToDevice, st));
  if (k->m) {
    KK_TRY(k, hipMemcpyAsync(k->s, s, (size_t)k->m * 8, hipMemcpyHostToDevice, st));
    KK_TRY(k, hipMemcpyAsync(k->y, y, (size_t)k->m * 8, hipMemcpyHostToDevice, st));
  }
  const size_t e1 = k->tm_form.mark(st);
  {
    // CSR-ordered value copies, Sigma = y ./ s, diag(H): everything the row-wise products and the assembly read
    const int64_t tot = std::max(std::max(k->nnzJ, k->nnzH), std::max(k->n, k->m));
    if (tot) hipLaunchKernelGGL(k_form_prep, grid1(tot), dim3(256), 0, st, k->nnzJ, k->nnzH, k->n, k->m, k->Jx, k->Jrmap, k->Hx, k->Hrmap, k->Hp, k->Hi,
                                k->s, k->y, k->Jcsr, k->Hcsr, k->Hdiag, k->sig);
  }
  if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC) {
    // form_system!(::Clever_Symmetric_KKT_solver), clever_symmetric.jl:341-393
    KK_TRY(k, hipMemsetAsync(k->Avals, 0, (size_t)std::max<int64_t>(k->nnzA, 1) * 8, st));
    const int64_t mn = k->m_new;
    if (mn) hipLaunchKernelGGL(k_clever_groups, grid1(mn), dim3(256), 0, st, mn, k->gptr, k->mind, k->mratio, k->s, k->y, k->gU, k->rowg);
    const double xscale = 1.0 / std::sqrt(1.0 + k->rescale_xinf);
    if (k->n + mn) hipLaunchKernelGGL(k_clever_rescale, grid1(k->n + mn), dim3(256), 0, st, k->n, mn, k->rescale_mode, k->rescale_mu, xscale, k->gU, k->Dres);
    const int64_t tot = k->nnzH + k->nnzJ + mn;
    if (tot) hipLaunchKernelGGL(k_clever_assemble, grid1(tot), dim3(256), 0, st, k->nnzH, k->nnzJ, k->n, mn, k->Hx, k->Jx, k->mapH, k->mapJc,
                                k->Hcol, k->Jcol, k->dAi, k->diagA, k->gU, k->Dres, k->Avals);
    if (k->n) {
      hipLaunchKernelGGL(k_clever_true_diag, grid1(k->n), dim3(256), 0, st, k->n, k->Hp, k->Hi, k->Hx, k->true_x_diag);
      SEG_LAUNCH(k_schur_diag_seg, k->lprJc, k->n, st, k->n, k->Jp, k->Ji, k->Jx, k->sig, k->Hdiag, k->schur_diag);
    }
  } else if (k->kind == OKKT_KKT_SYMMETRIC) {
    KK_TRY(k, hipMemsetAsync(k->Avals, 0, (size_t)std::max<int64_t>(k->nnzA, 1) * 8, st));
    const int64_t tot = k->nnzH + k->nnzJ + k->m;
    if (tot) hipLaunchKernelGGL(k_assemble_aug, grid1(tot), dim3(256), 0, st, k->nnzH, k->nnzJ, k->n, k->m, k->Hx, k->Jx, k->s, k->y, k->mapH, k->mapJ, k->diagA, k->Avals);
    SEG_LAUNCH(k_schur_diag_seg, k->lprJc, k->n, st, k->n, k->Jp, k->Ji, k->Jx, k->sig, k->Hdiag, k->schur_diag);
  } else {
    if (k->nnzA && k->schur_groups > 0) {
      const int G = k->schur_groups;
      hipLaunchKernelGGL(k_assemble_schur_lds, dim3((unsigned)((k->n + G - 1) / G)), dim3(16 * G), (size_t)G * k->schur_maxcol * sizeof(double), st, k->n, G,
                         k->schur_maxcol, k->dAp64, k->Jp, k->Ji, k->Jx, k->Jrp, k->Jcsr, k->seg_q, k->seg_t, k->tslot, k->sig, k->qh, k->Hx, k->Avals);
    } else if (k->nnzA) {
      hipLaunchKernelGGL(k_assemble_schur, grid1(k->nnzA), dim3(256), 0, st, k->nnzA, k->qptr, k->qa, k->qb, k->qi, k->qh, k->Jx, k->sig, k->Hx, k->Avals);
    }
    if (k->n) hipLaunchKernelGGL(k_gather, grid1(k->n), dim3(256), 0, st, k->n, k->diagA, k->Avals, k->schur_diag);   // schur_diag = diag(Q), schur.jl:56
  }
  const size_t e2 = k->tm_form.mark(st);
  k->tm_form.seg(0, e0, e1);
  k->tm_form.seg(1, e1, e2);
  KK_TRY(k, hipStreamSynchronize(st));
  KK_TRY(k, hipGetLastError());
  k->formed = true;
  k->factored = false;
  return OKKT_OK;
}

int okkt_kkt_diag_min(okkt_kkt_handle k, double* out) {
  if (!k || !out) return OKKT_ERR_INVALID;
  if (!k->formed) return kk_fail(k, OKKT_ERR_INVALID, "form_system has not been called");
  return kk_reduce(k, k->n, k->schur_diag, 0, out);
}

// is_diag_dom(kkt_solver.Q[1:n,1:n]) of delta_strategy.jl:1-9,95 at the delta of the last factor call, on the device: O(nnz)
// segmented sums instead of the reference's O(n nnz) sparse slicing.  out = 1 (dominant: the reference prints "Inertia
// calculation incorrect" when the inertia flag was 0), 0 (not), -1 (not evaluated: clever-symmetric system).
int okkt_kkt_is_diag_dom(okkt_kkt_handle k, int32_t* out) {
  if (!k || !out) return OKKT_ERR_INVALID;
  if (!k->formed) return kk_fail(k, OKKT_ERR_INVALID, "form_system has not been called");
  *out = -1;
  if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC) return OKKT_OK;
  const int64_t n = k->n, m = k->m;
  if (n == 0) { *out = 1; return OKKT_OK; }
  hipStream_t st = kk_stream(k);
  if (!k->ones) {
    const int64_t len = std::max(n, m);
    int rc2 = kk_alloc(k, (size_t)len, &k->ones);
    if (rc2 != OKKT_OK) return rc2;
    std::vector<double> one((size_t)len, 1.0);
    KK_TRY(k, hipMemcpyAsync(k->ones, one.data(), (size_t)len * 8, hipMemcpyHostToDevice, st));
    KK_TRY(k, hipStreamSynchronize(st));
  }
  const bool schur = k->kind == OKKT_KKT_SCHUR || k->kind == OKKT_KKT_SCHUR_DIRECT;
  // column sums of the stored H (CSC columns as rows of the segmented product), row sums (CSR copy)
  SEG_LAUNCH(k_seg_spmv, k->lprH, n, st, n, k->Hp, k->Hi, k->Hx, k->ones, (const double*)nullptr, (const double*)nullptr, 0.0, k->vn1);
  SEG_LAUNCH(k_seg_spmv, k->lprH, n, st, n, k->Hrp, k->Hrj, k->Hcsr, k->ones, (const double*)nullptr, (const double*)nullptr, 0.0, k->vn2);
  if (schur && m) {   // (J' Sigma J) 1 = J' (Sigma .* (J 1)): row sums = column sums (symmetric block, stored in full)
    SEG_LAUNCH(k_seg_spmv, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, k->ones, k->sig, (const double*)nullptr, 0.0, k->vm2);
    SEG_LAUNCH(k_seg_spmv, k->lprJc, n, st, n, k->Jp, k->Ji, k->Jx, k->vm2, (const double*)nullptr, (const double*)nullptr, 0.0, k->vn3);
  }
  hipLaunchKernelGGL(k_diag_dom_margin, grid1(n), dim3(256), 0, st, n, k->vn1, k->vn2, (schur && m) ? k->vn3 : (const double*)nullptr,
                     schur ? k->schur_diag : k->Hdiag, k->delta, k->big1);
  double mn = 0.0;
  int rc = kk_reduce(k, n, k->big1, 0, &mn);
  if (rc != OKKT_OK) return rc;
  *out = mn < 0.0 ? 0 : 1;
  return OKKT_OK;
}
int okkt_kkt_diag_dom_warnings(okkt_kkt_handle k, int32_t* count) {
  if (!k || !count) return OKKT_ERR_INVALID;
  *count = k->diag_dom_warnings;
  return OKKT_OK;
}

// estimate_y_tilde's tail (guess-vars.jl:155-160) on the device: dx = F \ (-g), y = -J dx with the factor and the Jacobian
// the handle holds (the Schur system of Sigma = I, H = lambda I, Cholesky semantics); nothing but g and y crosses PCIe
int okkt_kkt_estimate_y_tilde(okkt_kkt_handle k, const double* g, double* y_out) {
  if (!k || !g || !y_out) return OKKT_ERR_INVALID;
  if (k->kind != OKKT_KKT_SCHUR && k->kind != OKKT_KKT_SCHUR_DIRECT) return kk_fail(k, OKKT_ERR_INVALID, "estimate_y_tilde runs on the Schur system");
  if (!k->factored || !k->ls->factored) return kk_fail(k, OKKT_ERR_INVALID, "kkt solver not ready: factor! first");
  const int64_t n = k->n, m = k->m;
  hipStream_t st = kk_stream(k);
  if (n) {
    KK_TRY(k, hipMemcpyAsync(k->vn1, g, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_scale, grid1(n), dim3(256), 0, st, n, -1.0, k->vn1, k->vn2);
    int rc = solver_solve_enqueue(k->ls, k->vn2, k->dx, 1, false);
    if (rc != OKKT_OK) return kk_check_ls(k, rc, "ls_solve");
  }
  if (m) {
    SEG_LAUNCH(k_seg_spmv, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, k->dx, (const double*)nullptr, (const double*)nullptr, 0.0, k->vm1);
    hipLaunchKernelGGL(k_scale, grid1(m), dim3(256), 0, st, m, -1.0, k->vm1, k->vm2);
    KK_TRY(k, hipMemcpyAsync(y_out, k->vm2, (size_t)m * 8, hipMemcpyDeviceToHost, st));
  }
  KK_TRY(k, hipStreamSynchronize(st));
  KK_TRY(k, hipGetLastError());
  k->have_dir = false;
  return OKKT_OK;
}

// factor!(kkt_solver, delta): update_delta! then factor! (kkt_system_solver.jl:98-113,190-204); the state machine of the
// reference (:system_formed -> :delta_updated -> :factored) collapses into this one call.  trial: the caller throws a
// factorisation with the wrong inertia away (the delta loop), so it may stop as soon as the flag is decided.
static int kkt_factor_impl(okkt_kkt_s* k, double delta, okkt_inertia* inertia_out, bool trial) {
  if (!k->formed) return kk_fail(k, OKKT_ERR_INVALID, "kkt solver not ready to factor: form_system has not been called");
  int rc = solver_ensure_numeric(k->ls);
  if (rc != OKKT_OK) return kk_check_ls(k, rc, "device plan");
  static const bool env_early = !(getenv("OKKT_EARLY_EXIT") && atoi(getenv("OKKT_EARLY_EXIT")) == 0);
  const bool saved_early = k->ls->early_exit;
  if (trial && env_early) k->ls->early_exit = true;
  k->delta = delta;
  k->factored = false;
  k->have_dir = false;
  hipStream_t st = kk_stream(k);
  k->tm_factor.reset();
  const size_t e0 = k->tm_factor.mark(st);
  // the first n diagonal entries move, the (2,2) block never does (symmetric.jl:85-102)
  launch_set_shift(k->ls->N, delta, k->n);
  if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC && k->n)
    hipLaunchKernelGGL(k_clever_xdiag, grid1(k->n), dim3(256), 0, st, k->n, delta != 0.0 ? 1 : 0, k->Dres, k->true_x_diag, k->diagA, k->Avals);
  const size_t e1 = k->tm_factor.mark(st);
  k->tm_factor.seg(0, e0, e1);
  if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC) rc = solver_factor_device(k->ls, k->Avals, k->n, k->m_new, OKKT_SYM_SYMMETRIC, inertia_out);   // clever_symmetric.jl:395-400
  else if (k->kind == OKKT_KKT_SYMMETRIC) rc = solver_factor_device(k->ls, k->Avals, k->n, k->m, OKKT_SYM_SYMMETRIC, inertia_out);
  else rc = solver_factor_device(k->ls, k->Avals, k->n, 0, OKKT_SYM_DEFINITE, inertia_out);
  k->ls->early_exit = saved_early;
  k->t_factor_ms = k->ls->last_factor_ms;
  if (rc < 0) return kk_check_ls(k, rc, "factor");
  k->factored = true;
  return rc;
}

int okkt_kkt_factor(okkt_kkt_handle k, double delta, okkt_inertia* inertia_out) {
  if (!k) return OKKT_ERR_INVALID;
  return kkt_factor_impl(k, delta, inertia_out, false);
}
int okkt_kkt_factor_trial(okkt_kkt_handle k, double delta, okkt_inertia* inertia_out) {
  if (!k) return OKKT_ERR_INVALID;
  return kkt_factor_impl(k, delta, inertia_out, true);
}

int okkt_kkt_get_timers(okkt_kkt_handle k, okkt_kkt_timers* out) {
  if (!k || !out) return OKKT_ERR_INVALID;
  std::memset(out, 0, sizeof(*out));
  if (!k->structured) return kk_fail(k, OKKT_ERR_INVALID, "structure not set");
  KK_TRY(k, hipStreamSynchronize(kk_stream(k)));
  out->upload_ms = k->tm_form.sum(0);
  out->assemble_ms = k->tm_form.sum(1);
  out->shift_ms = k->tm_factor.sum(0);
  out->factor_ms = k->t_factor_ms;
  out->rhs_ms = k->tm_rhs.sum(0);
  out->solve_ms = k->tm_dir.sum(0);
  out->refine_ms = k->tm_dir.sum(1);
  out->kkt_err_ms = k->tm_dir.sum(2);
  out->direction_ms = k->tm_dir.sum(3);
  out->n_solves = k->n_solves;
  return OKKT_OK;
}

int okkt_kkt_ipopt_strategy(okkt_kkt_handle k, double delta_prev, const okkt_kkt_pars* pars, int32_t* num_fac_out, double* delta_out) {
  if (!k || !num_fac_out || !delta_out) return OKKT_ERR_INVALID;
  okkt_kkt_pars P;
  okkt_kkt_default_pars(&P);
  if (pars) P = *pars;
  // delta_strategy.jl:37-114
  int num_fac = 0;
  double dmin = 0.0;
  k->diag_dom_warnings = 0;
  // after a failed attempt the reference scans the x-block for diagonal dominance and prints a warning
  // (delta_strategy.jl:94-98); here a device scan, counted for the glue to print (okkt_kkt_diag_dom_warnings)
  auto scan_after_failure = [&]() -> int {
    static const bool on = !(getenv("OKKT_DIAG_DOM_SCAN") && atoi(getenv("OKKT_DIAG_DOM_SCAN")) == 0);
    if (!on) return OKKT_OK;
    int32_t dom = -1;
    const int rc3 = okkt_kkt_is_diag_dom(k, &dom);
    if (rc3 == OKKT_OK && dom == 1) ++k->diag_dom_warnings;
    return rc3;
  };
  int rc = okkt_kkt_diag_min(k, &dmin);
  if (rc != OKKT_OK) return rc;
  double tau = 1.5 * dmin;
  double delta = P.delta_zero;
  *num_fac_out = 0;
  *delta_out = delta;
  if (tau > 0.0) {
    tau = 0.0;
    rc = okkt_kkt_factor_trial(k, delta, nullptr);   // a failed attempt is discarded: it may stop early
    if (rc < 0) return rc;
    ++num_fac;
    if (rc == 1) { *num_fac_out = num_fac; *delta_out = delta; return 1; }
    if ((rc = scan_after_failure()) < 0) return rc;
  }
  for (int i = 1; i <= P.max_it; ++i) {
    if (i == 1) {
      if (delta_prev != 0.0) delta = std::max(P.delta_min - tau, delta_prev * P.delta_dec);
      else delta = P.delta_start - tau;
    } else {
      delta = delta * P.delta_inc;
    }
    rc = okkt_kkt_factor_trial(k, delta, nullptr);   // a failed attempt is discarded: it may stop early
    if (rc < 0) return rc;
    ++num_fac;
    *num_fac_out = num_fac;
    *delta_out = delta;
    if (rc == 1) return 1;
    { const int rc3 = scan_after_failure(); if (rc3 < 0) return rc3; }
    if (delta > P.delta_max) {   // :failure
      // The reference's default initialiser does not look at the status: gertz_init.jl:25-27 runs ipopt_strategy!,
      // kkt_associate_rhs! and compute_direction! whatever came back, i.e. it solves with the failed factor of the last
      // delta.  A trial that stopped early has no factor to solve with: complete it (not counted as an attempt).
      if (!k->ls->factored) {
        rc = okkt_kkt_factor(k, delta, nullptr);
        if (rc < 0) return rc;
      }
      return 0;
    }
  }
  return kk_fail(k, OKKT_ERR_INTERNAL, "max it");   // error("max it"), delta_strategy.jl:113
}

int okkt_kkt_system_rhs(okkt_kkt_handle k, const double* J_nzval_cur, const double* grad, const double* cons,
                        const double* s, const double* y, double mu, double a_norm_penalty,
                        double eta_P, double eta_D, double eta_mu, double* dual_r, double* primal_r, double* comp_r) {
  if (!k || !grad || !cons || !s || !y) return OKKT_ERR_INVALID;
  if (!k->structured) return kk_fail(k, OKKT_ERR_INVALID, "structure not set");
  if (!k->formed) return kk_fail(k, OKKT_ERR_INVALID, "form_system has not been called");
  hipStream_t st = kk_stream(k);
  const int64_t n = k->n, m = k->m;
  // J of the CURRENT iterate (may differ from the factorised one in correction steps, one_phase.jl:262-279)
  const double* Jx = k->Jx;
  k->cur_Jcsr = k->Jcsr;
  if (J_nzval_cur && k->nnzJ) {
    if (!k->Jcur) { int rc2 = kk_alloc(k, (size_t)k->nnzJ, &k->Jcur); if (rc2 != OKKT_OK) return rc2; }   // kept for the life of the handle
    KK_TRY(k, hipMemcpyAsync(k->Jcur, J_nzval_cur, (size_t)k->nnzJ * 8, hipMemcpyHostToDevice, st));
    Jx = k->Jcur;
    if (k->kind == OKKT_KKT_SCHUR_DIRECT) {     // eval_jac_prod(current_it, dir.x) needs the rows of that Jacobian
      if (!k->Jcur_csr) { int rc2 = kk_alloc(k, (size_t)k->nnzJ, &k->Jcur_csr); if (rc2 != OKKT_OK) return rc2; }
      hipLaunchKernelGGL(k_gather, grid1(k->nnzJ), dim3(256), 0, st, k->nnzJ, k->Jrmap, k->Jcur, k->Jcur_csr);
      k->cur_Jcsr = k->Jcur_csr;
    }
  }
  k->cur_Jx = Jx;
  if (!k->cur_grad) { int rc2 = kk_alloc(k, (size_t)std::max<int64_t>(n, 1), &k->cur_grad); if (rc2 != OKKT_OK) return rc2; }
  if (!k->cur_cons) { int rc2 = kk_alloc(k, (size_t)std::max<int64_t>(m, 1), &k->cur_cons); if (rc2 != OKKT_OK) return rc2; }
  k->cur_mu = mu; k->cur_pen = a_norm_penalty;
  if (n) {
    KK_TRY(k, hipMemcpyAsync(k->cur_grad, grad, (size_t)n * 8, hipMemcpyHostToDevice, st));     // kept for okkt_kkt_compute_directions
    KK_TRY(k, hipMemcpyAsync(k->vn1, k->cur_grad, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
  }
  if (m) {
    KK_TRY(k, hipMemcpyAsync(k->cur_cons, cons, (size_t)m * 8, hipMemcpyHostToDevice, st));
    KK_TRY(k, hipMemcpyAsync(k->vm1, k->cur_cons, (size_t)m * 8, hipMemcpyDeviceToDevice, st));
    KK_TRY(k, hipMemcpyAsync(k->cur_s, s, (size_t)m * 8, hipMemcpyHostToDevice, st));
    KK_TRY(k, hipMemcpyAsync(k->cur_y, y, (size_t)m * 8, hipMemcpyHostToDevice, st));
  }
  k->tm_rhs.reset();
  const size_t e0 = k->tm_rhs.mark(st);
  SEG_LAUNCH(k_rhs_dual_seg, k->lprJc, n, st, n, k->Jp, k->Ji, Jx, k->cur_y, k->vn1, (mu * eta_mu) * a_norm_penalty, 1.0 - eta_D, k->rD);
  if (m) {
    hipLaunchKernelGGL(k_rhs_pc, grid1(m), dim3(256), 0, st, m, k->vm1, k->cur_s, k->cur_y, 1.0 - eta_P, mu * eta_mu, k->rP, k->rC);
    hipLaunchKernelGGL(k_div, grid1(m), dim3(256), 0, st, m, k->cur_y, k->cur_s, k->cur_sig);   // Sigma of the current iterate (schur_direct.jl:47)
  }
  const size_t e1 = k->tm_rhs.mark(st);
  k->tm_rhs.seg(0, e0, e1);
  if (n && dual_r) KK_TRY(k, hipMemcpyAsync(dual_r, k->rD, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  if (m && primal_r) KK_TRY(k, hipMemcpyAsync(primal_r, k->rP, (size_t)m * 8, hipMemcpyDeviceToHost, st));
  if (m && comp_r) KK_TRY(k, hipMemcpyAsync(comp_r, k->rC, (size_t)m * 8, hipMemcpyDeviceToHost, st));
  KK_TRY(k, hipStreamSynchronize(st));
  KK_TRY(k, hipGetLastError());
  k->have_cur = true;
  k->have_rhs = true;
  return OKKT_OK;
}

int okkt_kkt_compute_direction(okkt_kkt_handle k, const double* dual_r, const double* primal_r, const double* comp_r,
                               int32_t ItRefine_Num, double* dx, double* dy, double* ds, okkt_kkt_error* err_out) {
  if (!k) return OKKT_ERR_INVALID;
  const bool host_rhs = dual_r || primal_r || comp_r;
  if (host_rhs && (!dual_r || !primal_r || !comp_r)) return kk_fail(k, OKKT_ERR_INVALID, "the rhs triple is given as three host vectors or as three NULLs (resident rhs)");
  const bool host_dir = dx || dy || ds;
  if (host_dir && (!dx || !dy || !ds)) return kk_fail(k, OKKT_ERR_INVALID, "dx, dy, ds: three host vectors or three NULLs");
  if (!k->factored) return kk_fail(k, OKKT_ERR_INVALID, "kkt solver not ready to compute direction!");  // kkt_system_solver.jl:181-183
  if (!k->ls->factored)
    return kk_fail(k, OKKT_ERR_INVALID, "the last factorisation was a discarded trial of the delta loop (it stopped early): factor! again before a direction");
  if (!host_rhs && !k->have_rhs) return kk_fail(k, OKKT_ERR_INVALID, "no resident rhs: okkt_kkt_system_rhs has not been called");
  const bool direct = k->kind == OKKT_KKT_SCHUR_DIRECT;
  if (direct && !k->have_cur) return kk_fail(k, OKKT_ERR_INVALID, "Schur_KKT_solver_direct reads current_it: kkt_associate_rhs! (okkt_kkt_system_rhs) has not been called");
  hipStream_t st = kk_stream(k);
  const int64_t n = k->n, m = k->m;
  std::string e;
  if (host_rhs) {
    if (n) KK_TRY(k, hipMemcpyAsync(k->rD, dual_r, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (m) {
      KK_TRY(k, hipMemcpyAsync(k->rP, primal_r, (size_t)m * 8, hipMemcpyHostToDevice, st));
      KK_TRY(k, hipMemcpyAsync(k->rC, comp_r, (size_t)m * 8, hipMemcpyHostToDevice, st));
    }
    k->have_rhs = true;
  }
  okkt_kkt_s::Timer& T = k->tm_dir;
  T.reset();
  k->n_solves = 0;
  const size_t t_begin = T.mark(st);
  size_t t_last = t_begin;
  auto lap = [&](int tag) { const size_t t = T.mark(st); T.seg(tag, t_last, t); t_last = t; };   // tags: 0 solve, 1 vector work, 2 N err
  auto solve = [&](const double* rhs, double* sol, bool accumulate) -> int {
    lap(1);
    const int rc2 = solver_solve_enqueue(k->ls, rhs, sol, 1, accumulate);
    lap(0);
    ++k->n_solves;
    return rc2;
  };
  int rc;
  if (k->kind == OKKT_KKT_SCHUR || direct) {
    // schur.jl:89-128 / schur_direct.jl:32-66 + solver_schur_rhs schur.jl:131-182.  The rhs terms, dy and ds take y, s, J of
    // factor_it (schur) or of current_it (direct); the refinement always works on the factorised system.
    const double* ys = direct ? k->cur_y : k->y;
    const double* ss = direct ? k->cur_s : k->s;
    const double* sg = direct ? k->cur_sig : k->sig;
    const double* Jc = direct ? k->cur_Jx : k->Jx;
    const double* Jr = direct ? k->cur_Jcsr : k->Jcsr;
    if (m) hipLaunchKernelGGL(k_schur_t1, grid1(m), dim3(256), 0, st, m, k->rP, k->rC, sg, ss, k->vm1);
    // vn2 = schur_rhs = dual_r + J' y_
    SEG_LAUNCH(k_seg_spmv, k->lprJc, n, st, n, k->Jp, k->Ji, Jc, k->vm1, (const double*)nullptr, k->rD, 1.0, k->vn2);
    if (n) KK_TRY(k, hipMemsetAsync(k->dx, 0, (size_t)n * 8, st));
    for (int it = 0; it < ItRefine_Num; ++it) {
      rc = solve(it == 0 ? k->vn2 : k->big1, k->dx, true);          // dir_x .+= ls_solve(res_old)
      if (rc != OKKT_OK) return kk_check_ls(k, rc, "ls_solve");
      // the residual behind the last solve is only printed by the reference (output_level >= 4): not evaluated
      if (it + 1 < ItRefine_Num && n) {
        SEG_LAUNCH(k_seg_spmv, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, k->dx, k->sig, (const double*)nullptr, 0.0, k->vm2);   // Sigma .* (J dx)
        SEG_LAUNCH(k_schur_resid, k->lprJc, n, st, n, k->Jp, k->Ji, k->Jx, k->vm2, k->Hrp, k->Hrj, k->Hcsr, k->Hp, k->Hi, k->Hx, k->Hdiag, k->dx, k->vn2,
                   k->delta, k->big1);
      }
    }
    SEG_LAUNCH(k_schur_dyds, k->lprJr, m, st, m, k->Jrp, k->Jrj, Jr, k->dx, k->rP, k->rC, ys, ss, sg, direct ? 1 : 0, k->dy, k->ds);
  } else if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC) {
    // compute_direction_implementation!(::Clever_Symmetric_KKT_solver), clever_symmetric.jl:417-492
    const int64_t mn = k->m_new, dim = n + mn;
    if (m) hipLaunchKernelGGL(k_clever_symrhs, grid1(m), dim3(256), 0, st, m, k->rP, k->rC, k->y, k->vm1);
    if (mn) hipLaunchKernelGGL(k_clever_crhs, grid1(mn), dim3(256), 0, st, mn, k->gptr, k->mind, k->rowg, k->vm1, k->crhs);
    if (dim) {
      hipLaunchKernelGGL(k_clever_rhs, grid1(dim), dim3(256), 0, st, n, mn, k->rD, k->crhs, k->Dres, k->big1);   // rescaled rhs
      KK_TRY(k, hipMemsetAsync(k->big2, 0, (size_t)dim * 8, st));                                                  // sol
    }
    for (int it = 0; it < ItRefine_Num; ++it) {       // ls_solve with refinement, clever_symmetric.jl:402-415
      if (it > 0 && dim) {
        hipLaunchKernelGGL(k_spmv_symlower, grid1(dim), dim3(256), 0, st, dim, k->dAp, k->dAi, k->Avals, k->Arp, k->Arj, k->Armap, k->big2, k->big4);
        hipLaunchKernelGGL(k_clever_res, grid1(dim), dim3(256), 0, st, dim, n, k->delta, k->big1, k->big4, k->big2, k->big3);
      }
      rc = solve(it == 0 ? k->big1 : k->big3, k->big2, true);       // sol += ls_solve(err)
      if (rc != OKKT_OK) return kk_check_ls(k, rc, "ls_solve");
    }
    if (dim) hipLaunchKernelGGL(k_clever_unscale, grid1(dim), dim3(256), 0, st, n, mn, k->big2, k->Dres, k->dx, k->big4);   // big4 = v
    if (m) {
      hipLaunchKernelGGL(k_clever_y, grid1(m), dim3(256), 0, st, m, k->row_grp, k->row_ratio, k->s, k->y, k->vm1, k->crhs, k->gU, k->big4, k->dy);
      SEG_LAUNCH(k_seg_spmv, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, k->dx, (const double*)nullptr, k->rP, -1.0, k->ds);   // J dx - primal_r
    }
  } else {
    // symmetric.jl:59-83
    if (n + m) hipLaunchKernelGGL(k_sym_rhs, grid1(n + m), dim3(256), 0, st, n, m, k->rD, k->rP, k->rC, k->y, k->big1);
    rc = solve(k->big1, k->big2, false);
    if (rc != OKKT_OK) return kk_check_ls(k, rc, "ls_solve");
    if (n + m) hipLaunchKernelGGL(k_sym_split, grid1(n + m), dim3(256), 0, st, n, m, k->big2, k->dx, k->dy);
    SEG_LAUNCH(k_seg_spmv, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, k->dx, (const double*)nullptr, k->rP, -1.0, k->ds);       // J dx - primal_r
  }
  lap(1);
  // update_kkt_error! (p = Inf), kkt_system_solver.jl:67-96: always with the matrices of factor_it
  okkt_kkt_error E;
  std::memset(&E, 0, sizeof(E));
  const int64_t nbD = n ? (int64_t)seg_grid(n, k->lprJc).x : 0, nbM = m ? (int64_t)seg_grid(m, k->lprJr).x : 0;
  SEG_LAUNCH(k_err_dual, k->lprJc, n, st, n, k->Jp, k->Ji, k->Jx, k->dy, k->Hrp, k->Hrj, k->Hcsr, k->Hp, k->Hi, k->Hx, k->Hdiag, k->dx, k->rD, k->delta, k->part);
  SEG_LAUNCH(k_err_pc, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, k->dx, k->ds, k->dy, k->s, k->y, k->rP, k->rC, k->part + nbD * 8);
  hipLaunchKernelGGL(k_err_final, dim3(1), dim3(256), 0, st, nbD, nbM, k->part, k->red);
  lap(2);
  T.seg(3, t_begin, t_last);
  double red[6] = {0, 0, 0, 0, 0, 0};
  KK_TRY(k, hipMemcpyAsync(red, k->red, sizeof(red), hipMemcpyDeviceToHost, st));
  if (host_dir) {
    if (n) KK_TRY(k, hipMemcpyAsync(dx, k->dx, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    if (m) {
      KK_TRY(k, hipMemcpyAsync(dy, k->dy, (size_t)m * 8, hipMemcpyDeviceToHost, st));
      KK_TRY(k, hipMemcpyAsync(ds, k->ds, (size_t)m * 8, hipMemcpyDeviceToHost, st));
    }
  }
  KK_TRY(k, hipStreamSynchronize(st));       // the only synchronisation of the call
  KK_TRY(k, hipGetLastError());
  auto mx = [](double a, double b) { return (a != a || b != b) ? NAN : std::max(a, b); };
  E.error_D = red[0]; E.error_P = red[2]; E.error_mu = red[3];
  E.overall = mx(mx(red[0], red[2]), red[3]);
  E.rhs_norm = mx(mx(red[1], red[4]), red[5]);
  E.ratio = E.overall / E.rhs_norm;
  if (err_out) *err_out = E;
  k->have_dir = true;
  k->have_dxnorm = false;
  return OKKT_OK;
}


// compute_direction! for several reduction-factor triples in ONE pass over the factor: the probe of the aggressive step
// (Reduct_affine, take_step.jl:2-3) and the candidates of take_step2! (take_step.jl:34-66) are right-hand sides of the same
// factorised system -- System_rhs (system_rhs.jl:57-73) is evaluated on the device for every triple from the gradient, constraint
// values, s and y that the last okkt_kkt_system_rhs left in HBM, the triangular solves carry up to four right-hand sides per
// sweep (okkt_solve(nrhs)), the Schur refinement rounds are batched the same way.  etas: nrhs x (eta_P, eta_D, eta_mu);
// dx: nrhs x n, dy / ds: nrhs x m (any of the three NULL: not downloaded); err: nrhs records or NULL.
int okkt_kkt_compute_directions(okkt_kkt_handle k, int32_t nrhs, const double* etas, int32_t ItRefine_Num,
                                double* dx, double* dy, double* ds, okkt_kkt_error* err_out) {
  if (!k || !etas || nrhs < 1 || nrhs > 16) return OKKT_ERR_INVALID;
  if (k->kind == OKKT_KKT_CLEVER_SYMMETRIC) return kk_fail(k, OKKT_ERR_INVALID, "batched directions: schur, schur_direct and symmetric systems");
  if (!k->factored) return kk_fail(k, OKKT_ERR_INVALID, "kkt solver not ready to compute direction!");
  if (!k->ls->factored) return kk_fail(k, OKKT_ERR_INVALID, "the last factorisation was a discarded trial of the delta loop: factor! again before a direction");
  if (!k->have_cur || !k->cur_grad) return kk_fail(k, OKKT_ERR_INVALID, "okkt_kkt_system_rhs (kkt_associate_rhs!) has not been called: no resident iterate");
  const bool schur = k->kind != OKKT_KKT_SYMMETRIC, direct = k->kind == OKKT_KKT_SCHUR_DIRECT;
  hipStream_t st = kk_stream(k);
  const int64_t n = k->n, m = k->m, dim = schur ? n : n + m;
  if (k->batch_cap < nrhs) {
    const size_t c = (size_t)nrhs;
    int rc2;
    if ((rc2 = kk_alloc(k, c * std::max<int64_t>(n, 1), &k->b_rD)) || (rc2 = kk_alloc(k, c * std::max<int64_t>(m, 1), &k->b_rP)) ||
        (rc2 = kk_alloc(k, c * std::max<int64_t>(m, 1), &k->b_rC)) || (rc2 = kk_alloc(k, c * (size_t)(n + m + 1), &k->b_rhs)) ||
        (rc2 = kk_alloc(k, c * (size_t)(n + m + 1), &k->b_sol)) || (rc2 = kk_alloc(k, c * std::max<int64_t>(n, 1), &k->b_res)) ||
        (rc2 = kk_alloc(k, c * std::max<int64_t>(n, 1), &k->b_dx)) || (rc2 = kk_alloc(k, c * std::max<int64_t>(m, 1), &k->b_dy)) ||
        (rc2 = kk_alloc(k, c * std::max<int64_t>(m, 1), &k->b_ds)) || (rc2 = kk_alloc(k, c * 8, &k->b_red)))
      return rc2;
    k->batch_cap = nrhs;     // the smaller buffers of an earlier call stay on the handle's allocation list until it is destroyed
  }
  okkt_kkt_s::Timer& T = k->tm_dir;
  T.reset();
  k->n_solves = 0;
  const size_t t_begin = T.mark(st);
  size_t t_last = t_begin;
  auto lap = [&](int tag) { const size_t t = T.mark(st); T.seg(tag, t_last, t); t_last = t; };
  const double* Jcur = k->cur_Jx;
  const double* ys = direct ? k->cur_y : k->y;
  const double* ss = direct ? k->cur_s : k->s;
  const double* sg = direct ? k->cur_sig : k->sig;
  const double* Jc = direct ? k->cur_Jx : k->Jx;
  const double* Jr = direct ? k->cur_Jcsr : k->Jcsr;
  // System_rhs for every triple, then the rhs of the linear system
  for (int q = 0; q < nrhs; ++q) {
    const double eP = etas[3 * q], eD = etas[3 * q + 1], eM = etas[3 * q + 2];
    double* rD = k->b_rD + (size_t)q * n; double* rP = k->b_rP + (size_t)q * m; double* rC = k->b_rC + (size_t)q * m;
    SEG_LAUNCH(k_rhs_dual_seg, k->lprJc, n, st, n, k->Jp, k->Ji, Jcur, k->cur_y, k->cur_grad, (k->cur_mu * eM) * k->cur_pen, 1.0 - eD, rD);
    if (m) hipLaunchKernelGGL(k_rhs_pc, grid1(m), dim3(256), 0, st, m, k->cur_cons, k->cur_s, k->cur_y, 1.0 - eP, k->cur_mu * eM, rP, rC);
    if (schur) {
      if (m) hipLaunchKernelGGL(k_schur_t1, grid1(m), dim3(256), 0, st, m, rP, rC, sg, ss, k->vm1);
      SEG_LAUNCH(k_seg_spmv, k->lprJc, n, st, n, k->Jp, k->Ji, Jc, k->vm1, (const double*)nullptr, rD, 1.0, k->b_rhs + (size_t)q * n);
    } else if (dim) {
      hipLaunchKernelGGL(k_sym_rhs, grid1(dim), dim3(256), 0, st, n, m, rD, rP, rC, k->y, k->b_rhs + (size_t)q * dim);
    }
  }
  lap(1);
  int rc;
  if (schur) {
    if (n) KK_TRY(k, hipMemsetAsync(k->b_dx, 0, (size_t)nrhs * n * 8, st));
    for (int it = 0; it < ItRefine_Num; ++it) {
      rc = solver_solve_enqueue(k->ls, it == 0 ? k->b_rhs : k->b_res, k->b_dx, nrhs, true);     // dir_x .+= ls_solve(res_old), all right-hand sides per sweep
      if (rc != OKKT_OK) return kk_check_ls(k, rc, "ls_solve");
      k->n_solves += nrhs;
      lap(0);
      if (it + 1 < ItRefine_Num && n)
        for (int q = 0; q < nrhs; ++q) {
          double* dxq = k->b_dx + (size_t)q * n;
          SEG_LAUNCH(k_seg_spmv, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, dxq, k->sig, (const double*)nullptr, 0.0, k->vm2);
          SEG_LAUNCH(k_schur_resid, k->lprJc, n, st, n, k->Jp, k->Ji, k->Jx, k->vm2, k->Hrp, k->Hrj, k->Hcsr, k->Hp, k->Hi, k->Hx, k->Hdiag, dxq,
                     k->b_rhs + (size_t)q * n, k->delta, k->b_res + (size_t)q * n);
        }
      lap(1);
    }
    for (int q = 0; q < nrhs; ++q)
      SEG_LAUNCH(k_schur_dyds, k->lprJr, m, st, m, k->Jrp, k->Jrj, Jr, k->b_dx + (size_t)q * n, k->b_rP + (size_t)q * m, k->b_rC + (size_t)q * m, ys, ss, sg,
                 direct ? 1 : 0, k->b_dy + (size_t)q * m, k->b_ds + (size_t)q * m);
  } else {
    rc = solver_solve_enqueue(k->ls, k->b_rhs, k->b_sol, nrhs, false);
    if (rc != OKKT_OK) return kk_check_ls(k, rc, "ls_solve");
    k->n_solves += nrhs;
    lap(0);
    for (int q = 0; q < nrhs; ++q) {
      if (dim) hipLaunchKernelGGL(k_sym_split, grid1(dim), dim3(256), 0, st, n, m, k->b_sol + (size_t)q * dim, k->b_dx + (size_t)q * n, k->b_dy + (size_t)q * m);
      SEG_LAUNCH(k_seg_spmv, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, k->b_dx + (size_t)q * n, (const double*)nullptr, k->b_rP + (size_t)q * m, -1.0,
                 k->b_ds + (size_t)q * m);
    }
  }
  lap(1);
  const int64_t nbD = n ? (int64_t)seg_grid(n, k->lprJc).x : 0, nbM = m ? (int64_t)seg_grid(m, k->lprJr).x : 0;
  for (int q = 0; q < nrhs; ++q) {
    const double* dxq = k->b_dx + (size_t)q * n; const double* dyq = k->b_dy + (size_t)q * m; const double* dsq = k->b_ds + (size_t)q * m;
    SEG_LAUNCH(k_err_dual, k->lprJc, n, st, n, k->Jp, k->Ji, k->Jx, dyq, k->Hrp, k->Hrj, k->Hcsr, k->Hp, k->Hi, k->Hx, k->Hdiag, dxq, k->b_rD + (size_t)q * n, k->delta, k->part);
    SEG_LAUNCH(k_err_pc, k->lprJr, m, st, m, k->Jrp, k->Jrj, k->Jcsr, dxq, dsq, dyq, k->s, k->y, k->b_rP + (size_t)q * m, k->b_rC + (size_t)q * m, k->part + nbD * 8);
    hipLaunchKernelGGL(k_err_final, dim3(1), dim3(256), 0, st, nbD, nbM, k->part, k->b_red + (size_t)q * 8);
  }
  lap(2);
  T.seg(3, t_begin, t_last);
  std::vector<double> red((size_t)nrhs * 8, 0.0);
  KK_TRY(k, hipMemcpyAsync(red.data(), k->b_red, red.size() * 8, hipMemcpyDeviceToHost, st));
  if (dx && n) KK_TRY(k, hipMemcpyAsync(dx, k->b_dx, (size_t)nrhs * n * 8, hipMemcpyDeviceToHost, st));
  if (dy && m) KK_TRY(k, hipMemcpyAsync(dy, k->b_dy, (size_t)nrhs * m * 8, hipMemcpyDeviceToHost, st));
  if (ds && m) KK_TRY(k, hipMemcpyAsync(ds, k->b_ds, (size_t)nrhs * m * 8, hipMemcpyDeviceToHost, st));
  KK_TRY(k, hipStreamSynchronize(st));
  KK_TRY(k, hipGetLastError());
  auto mx = [](double a, double b) { return (a != a || b != b) ? NAN : std::max(a, b); };
  for (int q = 0; q < nrhs && err_out; ++q) {
    const double* r = red.data() + (size_t)q * 8;
    okkt_kkt_error E;
    E.error_D = r[0]; E.error_P = r[2]; E.error_mu = r[3];
    E.overall = mx(mx(r[0], r[2]), r[3]);
    E.rhs_norm = mx(mx(r[1], r[4]), r[5]);
    E.ratio = E.overall / E.rhs_norm;
    err_out[q] = E;
  }
  k->have_dir = false;     // the resident single direction of okkt_kkt_compute_direction is not touched; the step-side kernels keep refusing until it is set
  return OKKT_OK;
}

int okkt_kkt_get_direction(okkt_kkt_handle k, double* dx, double* dy, double* ds) {
  if (!k) return OKKT_ERR_INVALID;
  if (!k->have_dir) return kk_fail(k, OKKT_ERR_INVALID, "no resident direction");
  hipStream_t st = kk_stream(k);
  if (k->n && dx) KK_TRY(k, hipMemcpyAsync(dx, k->dx, (size_t)k->n * 8, hipMemcpyDeviceToHost, st));
  if (k->m && dy) KK_TRY(k, hipMemcpyAsync(dy, k->dy, (size_t)k->m * 8, hipMemcpyDeviceToHost, st));
  if (k->m && ds) KK_TRY(k, hipMemcpyAsync(ds, k->ds, (size_t)k->m * 8, hipMemcpyDeviceToHost, st));
  KK_TRY(k, hipStreamSynchronize(st));
  return OKKT_OK;
}

int okkt_kkt_set_rescale(okkt_kkt_handle k, int mode, double mu, double x_norm_inf) {
  if (!k) return OKKT_ERR_INVALID;
  if (k->kind != OKKT_KKT_CLEVER_SYMMETRIC) return kk_fail(k, OKKT_ERR_INVALID, "kkt_system_rescale applies to the clever-symmetric solver only");
  if (mode != OKKT_RESCALE_NONE && mode != OKKT_RESCALE_U_ONLY && mode != OKKT_RESCALE_U_AND_X) return kk_fail(k, OKKT_ERR_INVALID, "unknown rescale mode");
  k->rescale_mode = mode; k->rescale_mu = mu; k->rescale_xinf = x_norm_inf;
  return OKKT_OK;
}

int okkt_kkt_compute_indicies(okkt_kkt_handle k, const double* J_nzval, int64_t* m_new_out) {
  if (!k) return OKKT_ERR_INVALID;
  if (k->kind != OKKT_KKT_CLEVER_SYMMETRIC) return kk_fail(k, OKKT_ERR_INVALID, "compute_indicies applies to the clever-symmetric solver only");
  if (!k->structured) return kk_fail(k, OKKT_ERR_INVALID, "okkt_kkt_set_structure has not been called");
  if (k->indexed) return kk_fail(k, OKKT_ERR_INVALID, "the parallel-row grouping is computed once per handle (clever_symmetric.jl:53-61)");
  if (k->nnzJ > 0 && !J_nzval) return OKKT_ERR_INVALID;
  try {
    const int64_t n = k->n, m = k->m;
    const std::vector<int64_t>& rp = k->h_Jrp;
    const std::vector<int>& rj = k->h_Jrj;
    // rows of J = columns of J_T: indices ascending, values in stored order; rescaled = divided by the first value
    std::vector<double> val(k->nnzJ), sc(k->nnzJ);
    for (int64_t i = 0; i < m; ++i)
      for (int64_t q = rp[i]; q < rp[i + 1]; ++q) { val[q] = J_nzval[k->h_Jrmap[q]]; sc[q] = val[q] / val[rp[i]]; }
    auto len = [&](int64_t i) { return rp[i + 1] - rp[i]; };
    // compare_columns on the rescaled matrix (clever_symmetric.jl:107-155): strict order "i before j"
    auto before = [&](int64_t i, int64_t j) -> bool {
      if (len(j) == 0) return false;
      if (len(i) == 0) return true;
      if (rj[rp[i]] != rj[rp[j]]) return rj[rp[j]] < rj[rp[i]];     // larger first index sorts first
      if (len(i) != len(j)) return len(i) < len(j);
      for (int64_t t = 0; t < len(i); ++t) if (rj[rp[i] + t] != rj[rp[j] + t]) return rj[rp[i] + t] < rj[rp[j] + t];
      for (int64_t t = 0; t < len(i); ++t) {
        if (sc[rp[i] + t] < sc[rp[j] + t]) return true;
        if (sc[rp[i] + t] > sc[rp[j] + t]) return false;
      }
      return i < j;
    };
    std::vector<int64_t> sorted_cols(m);
    std::iota(sorted_cols.begin(), sorted_cols.end(), (int64_t)0);
    std::sort(sorted_cols.begin(), sorted_cols.end(), before);
    // columns_are_same on the UNscaled matrix (clever_symmetric.jl:63-88)
    auto same = [&](int64_t i, int64_t j) -> bool {
      if (len(i) != len(j)) return false;
      for (int64_t t = 0; t < len(i); ++t) if (rj[rp[i] + t] != rj[rp[j] + t]) return false;
      if (len(i) == 0) return true;
      const double ratio = val[rp[i]] / val[rp[j]];
      double ss = 0.0;
      for (int64_t t = 0; t < len(i); ++t) { const double d = val[rp[i] + t] - val[rp[j] + t] * ratio; ss += d * d; }
      return std::sqrt(ss) < 1e-16;
    };
    struct Grp { int64_t first; std::vector<int64_t> ind; std::vector<double> ratio; };
    std::vector<Grp> groups;
    for (int64_t bp = 0; bp < m; ++bp) {
      const int64_t cur = sorted_cols[bp];
      if (bp == 0 || !same(sorted_cols[bp - 1], cur)) { groups.push_back(Grp{cur, {}, {}}); }
      Grp& g = groups.back();
      double ratio = 1.0;
      if (cur != g.first) {
        ratio = len(cur) > 0 ? val[rp[cur]] / val[rp[g.first]] : 1.0;
        if (ratio == 0.0 || !std::isfinite(ratio)) return kk_fail(k, OKKT_ERR_INVALID, "clever_symmetric.jl: ratio = 0, NaN or Inf between parallel rows");
      }
      g.ind.push_back(cur);
      g.ratio.push_back(ratio);
    }
    std::sort(groups.begin(), groups.end(), [](const Grp& a, const Grp& b) { return a.first < b.first; });
    const int64_t mn = (int64_t)groups.size();
    k->m_new = mn;
    k->h_first.resize(mn); k->h_gptr.assign(mn + 1, 0); k->h_mind.clear(); k->h_mratio.clear();
    std::vector<int> mind, row_grp(m, 0), grp_of_first_row(m, -1);
    std::vector<double> row_ratio(m, 1.0);
    for (int64_t g = 0; g < mn; ++g) {
      k->h_first[g] = groups[g].first;
      grp_of_first_row[groups[g].first] = (int)g;
      for (size_t t = 0; t < groups[g].ind.size(); ++t) {
        k->h_mind.push_back(groups[g].ind[t]); k->h_mratio.push_back(groups[g].ratio[t]);
        mind.push_back((int)groups[g].ind[t]);
        row_grp[groups[g].ind[t]] = (int)g; row_ratio[groups[g].ind[t]] = groups[g].ratio[t];
      }
      k->h_gptr[g + 1] = (int64_t)k->h_mind.size();
    }
    // ---- pattern of M = [[H 0];[J_new -U_new]] (lower), J_new = J[first rows, :] (clever_symmetric.jl:357-367)
    const int64_t dim = n + mn;
    k->dimA = dim;
    std::vector<int64_t> Ap(dim + 1, 0), Ai, mapH(k->nnzH), mapJc(k->nnzJ, -1), diagA(dim);
    for (int64_t j = 0; j < n; ++j) {
      bool has_diag = false;
      for (int64_t p = k->h_Hp[j]; p < k->h_Hp[j + 1]; ++p) has_diag |= k->h_Hi[p] == j;
      diagA[j] = (int64_t)Ai.size();
      if (!has_diag) Ai.push_back(j);
      for (int64_t p = k->h_Hp[j]; p < k->h_Hp[j + 1]; ++p) {
        if (k->h_Hi[p] == j) diagA[j] = (int64_t)Ai.size();
        mapH[p] = (int64_t)Ai.size();
        Ai.push_back(k->h_Hi[p]);
      }
      // rows of J that lead a group, in increasing group order (= increasing row order: groups are sorted by first)
      for (int64_t p = k->h_Jp[j]; p < k->h_Jp[j + 1]; ++p) {
        const int g = grp_of_first_row[k->h_Ji[p]];
        if (g >= 0) { mapJc[p] = (int64_t)Ai.size(); Ai.push_back(n + g); }
      }
      Ap[j + 1] = (int64_t)Ai.size();
    }
    for (int64_t g = 0; g < mn; ++g) { diagA[n + g] = (int64_t)Ai.size(); Ai.push_back(n + g); Ap[n + g + 1] = (int64_t)Ai.size(); }
    k->nnzA = (int64_t)Ai.size();
    k->Ap = Ap; k->Ai = Ai;
    // CSR view of the lower pattern for the symmetric product of the refinement (vector_product, eval.jl:221-230)
    std::vector<int64_t> Arp(dim + 1, 0), Armap(k->nnzA);
    std::vector<int> Arj(k->nnzA), dAi(k->nnzA);
    for (int64_t e = 0; e < k->nnzA; ++e) { dAi[e] = (int)Ai[e]; ++Arp[Ai[e] + 1]; }
    for (int64_t i = 0; i < dim; ++i) Arp[i + 1] += Arp[i];
    {
      std::vector<int64_t> fill(Arp.begin(), Arp.end() - 1);
      for (int64_t j = 0; j < dim; ++j)
        for (int64_t e = Ap[j]; e < Ap[j + 1]; ++e) { const int64_t q = fill[Ai[e]]++; Arj[q] = (int)j; Armap[q] = e; }
    }
    int rc;
#define UPL(dst, vec) if ((rc = kk_upload(k, vec, &k->dst)) != OKKT_OK) return rc
    UPL(gptr, k->h_gptr); UPL(mind, mind); UPL(mratio, k->h_mratio); UPL(row_grp, row_grp); UPL(row_ratio, row_ratio);
    UPL(mapH, mapH); UPL(mapJc, mapJc); UPL(diagA, diagA); UPL(dAp, Ap); UPL(dAi, dAi); UPL(Arp, Arp); UPL(Arj, Arj); UPL(Armap, Armap);
#undef UPL
    if ((rc = kk_alloc(k, (size_t)k->nnzA, &k->Avals))) return rc;
    rc = okkt_analyze(k->ls, k->dimA, k->Ap.data(), k->Ai.data(), 0);
    if (rc != OKKT_OK) return kk_check_ls(k, rc, "okkt_analyze");
    k->indexed = true;
    if (m_new_out) *m_new_out = mn;
    return OKKT_OK;
  } catch (const std::bad_alloc&) {
    return kk_fail(k, OKKT_ERR_ALLOC, "out of host memory in okkt_kkt_compute_indicies");
  } catch (...) {
    return kk_fail(k, OKKT_ERR_INTERNAL, "unexpected exception in okkt_kkt_compute_indicies");
  }
}

int okkt_kkt_get_indicies(okkt_kkt_handle k, int64_t* first_para_indicies, int64_t* group_ptr, int64_t* member_ind,
                          double* member_ratio, double* member_u, double* member_g, double* group_u) {
  if (!k) return OKKT_ERR_INVALID;
  if (!k->indexed) return kk_fail(k, OKKT_ERR_INVALID, "okkt_kkt_compute_indicies has not been called");
  if (first_para_indicies) std::copy(k->h_first.begin(), k->h_first.end(), first_para_indicies);
  if (group_ptr) std::copy(k->h_gptr.begin(), k->h_gptr.end(), group_ptr);
  if (member_ind) std::copy(k->h_mind.begin(), k->h_mind.end(), member_ind);
  if (member_ratio) std::copy(k->h_mratio.begin(), k->h_mratio.end(), member_ratio);
  if (member_u || member_g || group_u) {
    if (!k->formed) return kk_fail(k, OKKT_ERR_INVALID, "u and g exist after form_system (update_indicies!)");
    KK_TRY(k, hipStreamSynchronize(kk_stream(k)));
    const int64_t m = k->m;
    std::vector<double> s(m), y(m), rowg(m);
    if (m) {
      KK_TRY(k, hipMemcpy(s.data(), k->s, (size_t)m * 8, hipMemcpyDeviceToHost));
      KK_TRY(k, hipMemcpy(y.data(), k->y, (size_t)m * 8, hipMemcpyDeviceToHost));
      KK_TRY(k, hipMemcpy(rowg.data(), k->rowg, (size_t)m * 8, hipMemcpyDeviceToHost));
    }
    for (size_t t = 0; t < k->h_mind.size(); ++t) {
      const int64_t j = k->h_mind[t];
      if (member_u) member_u[t] = s[j] / y[j];
      if (member_g) member_g[t] = rowg[j];
    }
    if (group_u && k->m_new) KK_TRY(k, hipMemcpy(group_u, k->gU, (size_t)k->m_new * 8, hipMemcpyDeviceToHost));
  }
  return OKKT_OK;
}

int okkt_kkt_get_matrix(okkt_kkt_handle k, int64_t* dim_out, int64_t* nnz_out, int64_t* colptr_out, int64_t* rowval_out, double* nzval_out) {
  if (!k) return OKKT_ERR_INVALID;
  if (!k->structured) return kk_fail(k, OKKT_ERR_INVALID, "structure not set");
  if (dim_out) *dim_out = k->dimA;
  if (nnz_out) *nnz_out = k->nnzA;
  if (colptr_out) std::copy(k->Ap.begin(), k->Ap.end(), colptr_out);
  if (rowval_out) std::copy(k->Ai.begin(), k->Ai.end(), rowval_out);
  if (nzval_out) {
    if (!k->formed) return kk_fail(k, OKKT_ERR_INVALID, "form_system has not been called");
    KK_TRY(k, hipStreamSynchronize(kk_stream(k)));
    if (k->nnzA) KK_TRY(k, hipMemcpy(nzval_out, k->Avals, (size_t)k->nnzA * 8, hipMemcpyDeviceToHost));
  }
  return OKKT_OK;
}

int okkt_kkt_get_schur_diag(okkt_kkt_handle k, double* out) {
  if (!k || !out) return OKKT_ERR_INVALID;
  if (!k->formed) return kk_fail(k, OKKT_ERR_INVALID, "form_system has not been called");
  KK_TRY(k, hipStreamSynchronize(kk_stream(k)));
  if (k->n) KK_TRY(k, hipMemcpy(out, k->schur_diag, (size_t)k->n * 8, hipMemcpyDeviceToHost));
  return OKKT_OK;
}

}  // extern "C"
