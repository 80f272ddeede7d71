// Numeric supernodal multifrontal LDL^T (no pivoting) and the triangular solves, gfx950.
//
// Replaces, for the reference's hot path, CHOLMOD's numeric factorisation and solve that
// `cholesky(Symmetric(Q,:L))` / `ldlt(Symmetric(Q,:L))` / `F \ b` reach
// (/root/reference/src/linear_system_solvers/julia.jl:34,52,101,110).
//
// Fronts are dense f x f column-major buffers in one HBM arena (see symbolic.h).  The tree is
// processed level by level; inside a level
//   * fronts of order <= small_max are assembled, factored and written back by ONE workgroup
//     with the whole front resident in LDS (k_front_small),
//   * larger fronts are assembled by column blocks (k_big_assemble) and factored by a blocked
//     right-looking sweep: LDS diagonal block (k_big_diag), row-parallel triangular solve
//     (k_big_trsm) and an FP64-MFMA trailing update C -= W * L^T (k_big_syrk).
// Extend-add is deterministic: every destination entry is owned by exactly one workgroup that
// adds the children's contribution blocks in a fixed order -- no floating-point atomics.
#include "numeric.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace okkt {

typedef double d4_t __attribute__((ext_vector_type(4)));

#define OKKT_HIP_TRY(expr)                                                         \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess)                                                         \
      return std::string(#expr) + ": " + hipGetErrorString(e__);                   \
  } while (0)

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void classify_pivot(double d, double tol, unsigned& pos, unsigned& neg,
                                                unsigned& zer, unsigned& bad) {
  // julia.jl:72-78: pos = d > tol, neg = d < -tol, zero = rest; NaN/Inf counted apart
  if (isnan(d) || isinf(d)) ++bad;
  else if (d > tol) ++pos;
  else if (d < -tol) ++neg;
  else ++zer;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ void flush_counts(unsigned long long* counters, unsigned pos, unsigned neg,
                                             unsigned zer, unsigned bad) {
  pos = wave_sum(pos); neg = wave_sum(neg); zer = wave_sum(zer); bad = wave_sum(bad);
  if ((threadIdx.x & 63) == 0) {
    if (pos) atomicAdd(&counters[0], (unsigned long long)pos);
    if (neg) atomicAdd(&counters[1], (unsigned long long)neg);
    if (zer) atomicAdd(&counters[2], (unsigned long long)zer);
    if (bad) atomicAdd(&counters[3], (unsigned long long)bad);
  }
}

// Right-looking LDL^T of the leading npiv columns of an LDS-resident lower-triangular front.
// F is column-major with leading dimension ldf; wcol is scratch of length f.
template <int TPB>
__device__ __forceinline__ void ldl_partial_lds(double* F, double* wcol, int ldf, int f, int npiv) {
  constexpr int G = TPB / 32;
  const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
  for (int j = 0; j < npiv; ++j) {
    const double d = F[j + j * ldf];
    for (int i = j + 1 + tid; i < f; i += TPB) {
      const double w = F[i + j * ldf];
      wcol[i] = w;            // w_i = l_ij * d_j
      F[i + j * ldf] = w / d; // l_ij
    }
    __syncthreads();
    for (int c = j + 1 + grp; c < f; c += G) {
      const double wc = wcol[c];
      for (int i = c + lane; i < f; i += 32) F[i + c * ldf] -= F[i + j * ldf] * wc;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// small fronts: one workgroup, front resident in LDS
// ------------------------------------------------------------------------------------------
template <int TPB>
__global__ __launch_bounds__(TPB) void k_front_small(DevPlan P, const int* __restrict__ list, double tol) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int G = TPB / 32;
  const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
  const int s = list[blockIdx.x];
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int ldf = f | 1;
  double* F = sm;
  double* wcol = sm + (size_t)ldf * f;
  double* front = P.arena + P.front_pos[s];

  for (int i = tid; i < ldf * f; i += TPB) F[i] = 0.0;
  __syncthreads();
  // original matrix entries of this supernode's columns
  {
    const int64_t e0 = P.aent_ptr[s], e1 = P.aent_ptr[s + 1];
    if (!P.has_dup) {
      for (int64_t e = e0 + tid; e < e1; e += TPB) {
        const int dst = P.aent_dst[e];
        const int lc = dst / f, lr = dst - lc * f;
        F[lr + lc * ldf] = P.vals[P.aent_src[e]];
      }
    } else if (tid == 0) {
      for (int64_t e = e0; e < e1; ++e) {
        const int dst = P.aent_dst[e];
        const int lc = dst / f, lr = dst - lc * f;
        F[lr + lc * ldf] += P.vals[P.aent_src[e]];
      }
    }
  }
  __syncthreads();
  for (int j = tid; j < k; j += TPB) F[j + j * ldf] += P.diagadd[col0 + j];
  __syncthreads();
  // extend-add of the children's contribution blocks, one child at a time (fixed order)
  for (int64_t q = P.child_ptr[s]; q < P.child_ptr[s + 1]; ++q) {
    const int c = P.children[q];
    const int kc = P.sn_col0[c + 1] - P.sn_col0[c];
    const int fc = (int)(P.row_ptr[c + 1] - P.row_ptr[c]);
    const int rc = fc - kc;
    const double* C = P.arena + P.front_pos[c];
    const int* rl = P.rel + P.rel_ptr[c];
    for (int jj = grp; jj < rc; jj += G) {
      const int pj = rl[jj];
      const double* Ccol = C + (size_t)(kc + jj) * fc + kc;
      for (int ii = jj + lane; ii < rc; ii += 32) F[rl[ii] + pj * ldf] += Ccol[ii];
    }
    __syncthreads();
  }
  ldl_partial_lds<TPB>(F, wcol, ldf, f, k);
  // write back: L panel (rows >= col), contribution block (lower), D, inertia counts
  for (int c = grp; c < f; c += G) {
    double* dst = front + (size_t)c * f;
    for (int i = c + lane; i < f; i += 32) dst[i] = F[i + c * ldf];
  }
  unsigned pos = 0, neg = 0, zer = 0, bad = 0;
  for (int j = tid; j < k; j += TPB) {
    const double d = F[j + j * ldf];
    P.dvals[col0 + j] = d;
    classify_pivot(d, tol, pos, neg, zer, bad);
  }
  flush_counts(P.counters, pos, neg, zer, bad);
}

// ------------------------------------------------------------------------------------------
// big fronts
// ------------------------------------------------------------------------------------------
constexpr int kAsmCols = 16;  // columns of the parent owned by one assemble workgroup

__device__ __forceinline__ int lower_bound_dev(const int* a, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

__global__ __launch_bounds__(256) void k_big_assemble(DevPlan P, const int* __restrict__ list) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int s = list[blockIdx.y];
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int c0 = blockIdx.x * kAsmCols;
  if (c0 >= f) return;
  const int c1 = min(c0 + kAsmCols, f);
  double* F = P.arena + P.front_pos[s];
  for (int c = c0 + wv; c < c1; c += 4) {
    double* col = F + (size_t)c * f;
    for (int i = c + lane; i < f; i += 64) col[i] = 0.0;
  }
  __syncthreads();
  if (c0 < k) {
    const int64_t e0 = P.aent_ptr[s];
    const int ne = (int)(P.aent_ptr[s + 1] - e0);
    const int* dstv = P.aent_dst + e0;
    const int lo = lower_bound_dev(dstv, ne, c0 * f);
    const int hi = lower_bound_dev(dstv, ne, c1 * f);  // c1*f <= f*f fits: f <= 46340 checked on host
    if (!P.has_dup) {
      for (int e = lo + tid; e < hi; e += 256) F[dstv[e]] = P.vals[P.aent_src[e0 + e]];
    } else if (tid == 0) {
      for (int e = lo; e < hi; ++e) F[dstv[e]] += P.vals[P.aent_src[e0 + e]];
    }
    __syncthreads();
    for (int c = c0 + tid; c < min(c1, k); c += 256) F[c + (size_t)c * f] += P.diagadd[col0 + c];
  }
  __syncthreads();
  for (int64_t q = P.child_ptr[s]; q < P.child_ptr[s + 1]; ++q) {
    const int c = P.children[q];
    const int kc = P.sn_col0[c + 1] - P.sn_col0[c];
    const int fc = (int)(P.row_ptr[c + 1] - P.row_ptr[c]);
    const int rc = fc - kc;
    const double* C = P.arena + P.front_pos[c];
    const int* rl = P.rel + P.rel_ptr[c];
    const int j_lo = lower_bound_dev(rl, rc, c0);
    const int j_hi = lower_bound_dev(rl, rc, c1);
    for (int jj = j_lo + wv; jj < j_hi; jj += 4) {
      double* pcol = F + (size_t)rl[jj] * f;
      const double* Ccol = C + (size_t)(kc + jj) * fc + kc;
      for (int ii = jj + lane; ii < rc; ii += 64) pcol[rl[ii]] += Ccol[ii];
    }
    __syncthreads();
  }
}

// diagonal block of block-column `step`: LDL^T in LDS, D and inertia out
__global__ __launch_bounds__(256) void k_big_diag(DevPlan P, const int* __restrict__ list, int step, int NB, double tol) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
  const int s = list[blockIdx.x];
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = step * NB;
  if (j0 >= k) return;
  const int nb = min(NB, k - j0);
  const int ldb = nb | 1;
  double* B = sm;
  double* wcol = sm + (size_t)ldb * nb;
  double* F = P.arena + P.front_pos[s];
  for (int c = grp; c < nb; c += 8) {
    const double* src = F + (size_t)(j0 + c) * f + j0;
    for (int i = c + lane; i < nb; i += 32) B[i + c * ldb] = src[i];
  }
  __syncthreads();
  ldl_partial_lds<256>(B, wcol, ldb, nb, nb);
  for (int c = grp; c < nb; c += 8) {
    double* dst = F + (size_t)(j0 + c) * f + j0;
    for (int i = c + lane; i < nb; i += 32) dst[i] = B[i + c * ldb];
  }
  unsigned pos = 0, neg = 0, zer = 0, bad = 0;
  for (int j = tid; j < nb; j += 256) {
    const double d = B[j + j * ldb];
    P.dvals[col0 + j0 + j] = d;
    classify_pivot(d, tol, pos, neg, zer, bad);
  }
  flush_counts(P.counters, pos, neg, zer, bad);
}

// rows below the diagonal block: W = A21 * L11^-T (kept for the trailing update), L21 = W * D^-1
constexpr int kTrsmRows = 64;
__global__ __launch_bounds__(kTrsmRows) void k_big_trsm(DevPlan P, const int* __restrict__ list, int step, int NB) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x;
  const int s = list[blockIdx.y];
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = step * NB;
  if (j0 >= k) return;
  const int nb = min(NB, k - j0);
  const int r0 = j0 + nb + blockIdx.x * kTrsmRows;
  if (r0 >= f) return;
  const int ldb = nb | 1;
  double* L11 = sm;                         // nb x nb (lower, unit diagonal implied)
  double* dd = sm + (size_t)ldb * nb;       // nb
  double* wl = dd + nb;                     // nb x kTrsmRows, wl[c*64 + tid]
  double* F = P.arena + P.front_pos[s];
  for (int idx = tid; idx < nb * nb; idx += kTrsmRows) {
    const int c = idx / nb, i = idx - c * nb;
    if (i >= c) L11[i + c * ldb] = F[(size_t)(j0 + c) * f + j0 + i];
  }
  __syncthreads();
  for (int j = tid; j < nb; j += kTrsmRows) dd[j] = L11[j + j * ldb];
  __syncthreads();
  const int i = r0 + tid;
  if (i < f) {
    double* Wb = P.wbuf + P.wbuf_pos[s];
    for (int c = 0; c < nb; ++c) {
      double v = F[(size_t)(j0 + c) * f + i];
      for (int p = 0; p < c; ++p) v -= wl[p * kTrsmRows + tid] * L11[c + p * ldb];
      wl[c * kTrsmRows + tid] = v;
    }
    for (int c = 0; c < nb; ++c) {
      const double w = wl[c * kTrsmRows + tid];
      Wb[(size_t)c * f + i] = w;
      F[(size_t)(j0 + c) * f + i] = w / dd[c];
    }
  }
}

// trailing update with FP64 MFMA:  F[r, c] -= sum_p W[r, p] * L[c, p]   for r >= c >= j0 + nb
// 64 x 64 tile per workgroup, 32 x 32 per wave, v_mfma_f64_16x16x4_f64.
// The product is formed transposed (D[c][r]) so that the 16 lanes of an MFMA row group hold 16
// consecutive front rows: loads and stores of F touch whole 128-byte segments.
__global__ __launch_bounds__(256) void k_big_syrk(DevPlan P, const int* __restrict__ list, int step, int NB) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int s = list[blockIdx.y];
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = step * NB;
  if (j0 >= k) return;
  const int nb = min(NB, k - j0);
  const int t0 = j0 + nb;
  const int T = (f - t0 + 63) >> 6;
  // tile pair (ti >= tj) from the flat index
  const int idx = blockIdx.x;
  if (idx >= T * (T + 1) / 2) return;
  int ti = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= idx) ++ti;
  while (ti * (ti + 1) / 2 > idx) --ti;
  const int tj = idx - ti * (ti + 1) / 2;
  const int rbase = t0 + ti * 64 + (wv & 1) * 32;  // front rows of this wave's 32 x 32 piece
  const int cbase = t0 + tj * 64 + (wv >> 1) * 32; // front columns
  if (rbase + 31 < cbase) return;                  // strictly above the diagonal
  if (rbase >= f || cbase >= f) return;
  double* F = P.arena + P.front_pos[s];
  const double* Wb = P.wbuf + P.wbuf_pos[s];
  const int l15 = lane & 15, l4 = lane >> 4;
  d4_t acc[2][2];  // [column block][row block]
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};
  const int cA0 = cbase + l15, cA1 = cbase + 16 + l15;
  const int rB0 = rbase + l15, rB1 = rbase + 16 + l15;
  for (int p0 = 0; p0 < nb; p0 += 4) {
    const int p = p0 + l4;
    const bool pv = p < nb;
    // A operand: A[i = c][kk = p] = L[c, j0 + p];  B operand: B[kk = p][j = r] = W[r, p]
    const double a0 = (pv && cA0 < f) ? F[(size_t)(j0 + p) * f + cA0] : 0.0;
    const double a1 = (pv && cA1 < f) ? F[(size_t)(j0 + p) * f + cA1] : 0.0;
    const double b0 = (pv && rB0 < f) ? Wb[(size_t)p * f + rB0] : 0.0;
    const double b1 = (pv && rB1 < f) ? Wb[(size_t)p * f + rB1] : 0.0;
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
  // D layout (f64 16x16x4): D[i = (lane>>4) + 4*reg][j = lane&15]  ->  i = column offset, j = row offset
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int r = rbase + b * 16 + l15;
      for (int reg = 0; reg < 4; ++reg) {
        const int c = cbase + a * 16 + l4 + 4 * reg;
        if (r < f && c < f && r >= c) F[(size_t)c * f + r] -= acc[a][b][reg];
      }
    }
}

// ------------------------------------------------------------------------------------------
// triangular solves (one workgroup per front and level)
// ------------------------------------------------------------------------------------------
template <int TPB>
__global__ __launch_bounds__(TPB) void k_solve_fwd(DevPlan P, const int* __restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x;
  const int s = list[blockIdx.x];
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const double* L = P.arena + P.front_pos[s];
  double* w = sm;
  for (int i = tid; i < f; i += TPB) w[i] = i < k ? P.xwork[col0 + i] : 0.0;
  __syncthreads();
  for (int64_t q = P.child_ptr[s]; q < P.child_ptr[s + 1]; ++q) {
    const int c = P.children[q];
    const int rc = (int)(P.rel_ptr[c + 1] - P.rel_ptr[c]);
    const int* rl = P.rel + P.rel_ptr[c];
    const double* cvc = P.cv + P.cv_pos[c];
    for (int ii = tid; ii < rc; ii += TPB) w[rl[ii]] += cvc[ii];
    __syncthreads();
  }
  // unit lower triangular k x k
  for (int j = 0; j < k; ++j) {
    const double yj = w[j];
    const double* col = L + (size_t)j * f;
    for (int i = j + 1 + tid; i < k; i += TPB) w[i] -= col[i] * yj;
    __syncthreads();
  }
  // rows below the pivot block: contribution vector for the parent
  double* cvs = P.cv + P.cv_pos[s];
  for (int i = k + tid; i < f; i += TPB) {
    double acc = w[i];
    for (int j = 0; j < k; ++j) acc -= L[(size_t)j * f + i] * w[j];
    cvs[i - k] = acc;
  }
  // z = D^-1 y
  for (int j = tid; j < k; j += TPB) P.xwork[col0 + j] = w[j] / P.dvals[col0 + j];
}

template <int TPB>
__global__ __launch_bounds__(TPB) void k_solve_bwd(DevPlan P, const int* __restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = TPB / 64;
  const int s = list[blockIdx.x];
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const double* L = P.arena + P.front_pos[s];
  const int* rows = P.rows + P.row_ptr[s];
  double* xo = sm;  // [f]: first k = rhs of the triangular solve, rest = ancestors' solution
  for (int i = tid; i < f; i += TPB) xo[i] = P.xwork[rows[i]];
  __syncthreads();
  // rhs_j = z_j - sum_{i >= k} L[i, j] * x_i
  for (int j = wv; j < k; j += NW) {
    const double* col = L + (size_t)j * f;
    double acc = 0.0;
    for (int i = k + lane; i < f; i += 64) acc += col[i] * xo[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) xo[j] -= acc;
  }
  __syncthreads();
  // unit upper triangular (L11^T) k x k, column oriented
  for (int j = k - 1; j >= 0; --j) {
    const double xj = xo[j];
    for (int i = tid; i < j; i += TPB) xo[i] -= L[(size_t)i * f + j] * xj;
    __syncthreads();
  }
  for (int j = tid; j < k; j += TPB) P.xwork[col0 + j] = xo[j];
}

__global__ void k_permute_in(int n, const int* __restrict__ perm, const double* __restrict__ rhs, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = rhs[perm[i]];
}
__global__ void k_permute_out(int n, const int* __restrict__ perm, const double* __restrict__ x, double* __restrict__ sol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sol[perm[i]] = x[i];
}
__global__ void k_set_shift(int n, const int* __restrict__ perm, double delta, int nshift, double* __restrict__ diagadd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) diagadd[i] = perm[i] < nshift ? delta : 0.0;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
namespace {

template <typename T>
std::string upload(Numeric& N, const std::vector<T>& v, T** out) {
  *out = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  void* p = nullptr;
  OKKT_HIP_TRY(hipMalloc(&p, bytes));
  N.allocations.push_back(p);
  if (!v.empty()) OKKT_HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return "";
}

template <typename T>
std::string dalloc(Numeric& N, size_t count, T** out, bool zero) {
  *out = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  void* p = nullptr;
  OKKT_HIP_TRY(hipMalloc(&p, bytes));
  N.allocations.push_back(p);
  if (zero) OKKT_HIP_TRY(hipMemset(p, 0, bytes));
  *out = (T*)p;
  return "";
}

size_t lds_small(int maxf) { return ((size_t)(maxf | 1) * maxf + maxf) * sizeof(double); }

}  // namespace

std::string numeric_setup(const Symbolic& S, const SymbolicOptions& opts, hipStream_t stream, Numeric& N) {
  N.stream = stream;
  N.nb = std::max(8, std::min(opts.panel_nb, 128));
  N.small_max = std::max(32, std::min(opts.small_front_max, 136));
  N.nnz_in = S.nnz_in;
  if (S.max_front > 46000) return "front order exceeds the 32-bit local offset range";
  const int ns = S.nsuper;
  DevPlan& d = N.d;
  d.n = (int)S.n;
  d.nsuper = ns;
  d.has_dup = S.has_duplicates ? 1 : 0;
  std::string e;
#define UP(field, vec) if (!(e = upload(N, vec, &d.field)).empty()) return e
  UP(sn_col0, S.sn_col0);
  UP(row_ptr, S.row_ptr);
  UP(rows, S.rows);
  UP(front_pos, S.front_pos);
  UP(child_ptr, S.child_ptr);
  UP(children, S.children);
  UP(rel_ptr, S.rel_ptr);
  UP(rel, S.rel);
  UP(cv_pos, S.cv_pos);
  UP(aent_ptr, S.aent_ptr);
  UP(aent_src, S.aent_src);
  UP(aent_dst, S.aent_dst);
  UP(perm, S.perm);
#undef UP
  // schedule: per level, fronts grouped by class, larger fronts first inside a class
  std::vector<int> sched;
  sched.reserve(ns);
  std::vector<int64_t> wpos(ns, -1);
  int64_t wtotal = 0;
  N.levels.assign(S.nlevels, LevelSchedule());
  N.n_small = N.n_big = 0;
  for (int l = 0; l < S.nlevels; ++l) {
    std::vector<int> cls[kNumClasses];
    for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) {
      int s = S.level_sn[q];
      int f = (int)(S.row_ptr[s + 1] - S.row_ptr[s]);
      int c = f <= 32 ? 0 : (f <= 64 ? 1 : (f <= N.small_max ? 2 : 3));
      cls[c].push_back(s);
    }
    for (int c = 0; c < kNumClasses; ++c) {
      auto& v = cls[c];
      std::stable_sort(v.begin(), v.end(), [&](int a, int b) {
        return (S.row_ptr[a + 1] - S.row_ptr[a]) > (S.row_ptr[b + 1] - S.row_ptr[b]);
      });
      Segment& g = N.levels[l].seg[c];
      g.off = (int)sched.size();
      g.cnt = (int)v.size();
      for (int s : v) {
        int f = (int)(S.row_ptr[s + 1] - S.row_ptr[s]);
        int k = S.sn_col0[s + 1] - S.sn_col0[s];
        g.maxf = std::max(g.maxf, f);
        g.maxk = std::max(g.maxk, k);
        sched.push_back(s);
        if (c == 3) { wpos[s] = wtotal; wtotal += (int64_t)f * N.nb; ++N.n_big; } else ++N.n_small;
      }
    }
  }
  if (!(e = upload(N, sched, &d.sched)).empty()) return e;
  if (!(e = upload(N, wpos, &d.wbuf_pos)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.arena_doubles, &d.arena, false)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.n, &d.dvals, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.n, &d.diagadd, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.n, &d.xwork, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.sum_r, &d.cv, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)wtotal, &d.wbuf, false)).empty()) return e;
  if (!(e = dalloc(N, (size_t)4, &d.counters, true)).empty()) return e;
  if (!(e = dalloc(N, (size_t)S.nnz_in, &N.vals_owned, false)).empty()) return e;
  // kernels that may want more than 64 KiB of dynamic LDS
  const int big_lds = 160 * 1024;
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_front_small<256>, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_big_diag, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  OKKT_HIP_TRY(hipFuncSetAttribute((const void*)k_big_trsm, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds));
  return "";
}

void numeric_release(Numeric& N) {
  for (void* p : N.allocations) (void)hipFree(p);
  N.allocations.clear();
  N.levels.clear();
  N.d = DevPlan();
  N.vals_owned = nullptr;
}

std::string numeric_factor_enqueue(Numeric& N, const double* d_vals, double tol) {
  DevPlan P = N.d;
  P.vals = d_vals;
  hipStream_t st = N.stream;
  OKKT_HIP_TRY(hipMemsetAsync(P.counters, 0, 4 * sizeof(unsigned long long), st));
  const int NB = N.nb;
  for (size_t l = 0; l < N.levels.size(); ++l) {
    const LevelSchedule& L = N.levels[l];
    if (L.seg[0].cnt) {
      const Segment& g = L.seg[0];
      hipLaunchKernelGGL(k_front_small<64>, dim3(g.cnt), dim3(64), lds_small(g.maxf), st, P, P.sched + g.off, tol);
    }
    for (int c = 1; c <= 2; ++c)
      if (L.seg[c].cnt) {
        const Segment& g = L.seg[c];
        hipLaunchKernelGGL(k_front_small<256>, dim3(g.cnt), dim3(256), lds_small(g.maxf), st, P, P.sched + g.off, tol);
      }
    if (L.seg[3].cnt) {
      const Segment& g = L.seg[3];
      const int* list = P.sched + g.off;
      hipLaunchKernelGGL(k_big_assemble, dim3((g.maxf + kAsmCols - 1) / kAsmCols, g.cnt), dim3(256), 0, st, P, list);
      const int nsteps = (g.maxk + NB - 1) / NB;
      const size_t lds_diag = ((size_t)(NB | 1) * NB + NB) * sizeof(double);
      const size_t lds_trsm = ((size_t)(NB | 1) * NB + NB + (size_t)NB * kTrsmRows) * sizeof(double);
      for (int step = 0; step < nsteps; ++step) {
        hipLaunchKernelGGL(k_big_diag, dim3(g.cnt), dim3(256), lds_diag, st, P, list, step, NB, tol);
        const int rem = g.maxf - step * NB;  // upper bound on rows below the diagonal block
        if (rem <= 0) continue;
        hipLaunchKernelGGL(k_big_trsm, dim3((rem + kTrsmRows - 1) / kTrsmRows, g.cnt), dim3(kTrsmRows), lds_trsm, st, P, list, step, NB);
        const int T = (rem + 63) / 64;
        hipLaunchKernelGGL(k_big_syrk, dim3(T * (T + 1) / 2, g.cnt), dim3(256), 0, st, P, list, step, NB);
      }
    }
  }
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}

std::string numeric_solve_enqueue(Numeric& N) {
  DevPlan P = N.d;
  hipStream_t st = N.stream;
  const int nl = (int)N.levels.size();
  for (int l = 0; l < nl; ++l) {
    const LevelSchedule& L = N.levels[l];
    for (int c = 0; c < kNumClasses; ++c) {
      const Segment& g = L.seg[c];
      if (!g.cnt) continue;
      const size_t lds = (size_t)g.maxf * sizeof(double);
      if (c == 0) hipLaunchKernelGGL(k_solve_fwd<64>, dim3(g.cnt), dim3(64), lds, st, P, P.sched + g.off);
      else if (c < 3) hipLaunchKernelGGL(k_solve_fwd<256>, dim3(g.cnt), dim3(256), lds, st, P, P.sched + g.off);
      else hipLaunchKernelGGL(k_solve_fwd<1024>, dim3(g.cnt), dim3(1024), lds, st, P, P.sched + g.off);
    }
  }
  for (int l = nl - 1; l >= 0; --l) {
    const LevelSchedule& L = N.levels[l];
    for (int c = 0; c < kNumClasses; ++c) {
      const Segment& g = L.seg[c];
      if (!g.cnt) continue;
      const size_t lds = (size_t)g.maxf * sizeof(double);
      if (c == 0) hipLaunchKernelGGL(k_solve_bwd<64>, dim3(g.cnt), dim3(64), lds, st, P, P.sched + g.off);
      else if (c < 3) hipLaunchKernelGGL(k_solve_bwd<256>, dim3(g.cnt), dim3(256), lds, st, P, P.sched + g.off);
      else hipLaunchKernelGGL(k_solve_bwd<1024>, dim3(g.cnt), dim3(1024), lds, st, P, P.sched + g.off);
    }
  }
  OKKT_HIP_TRY(hipGetLastError());
  return "";
}

void launch_permute_in(const Numeric& N, const double* d_rhs) {
  const int n = N.d.n;
  if (n) hipLaunchKernelGGL(k_permute_in, dim3((n + 255) / 256), dim3(256), 0, N.stream, n, N.d.perm, d_rhs, N.d.xwork);
}
void launch_permute_out(const Numeric& N, double* d_sol) {
  const int n = N.d.n;
  if (n) hipLaunchKernelGGL(k_permute_out, dim3((n + 255) / 256), dim3(256), 0, N.stream, n, N.d.perm, N.d.xwork, d_sol);
}
void launch_set_shift(const Numeric& N, double delta, int64_t nshift) {
  const int n = N.d.n;
  if (n) hipLaunchKernelGGL(k_set_shift, dim3((n + 255) / 256), dim3(256), 0, N.stream, n, N.d.perm, delta, (int)nshift, N.d.diagadd);
}

}  // namespace okkt
