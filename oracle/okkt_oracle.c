/*
 * okkt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product path (onephase.jl_amd/csrc) never links it and has no CPU fallback.
 *
 * What it restates.  The reference's linear solver is
 *     /root/reference/src/linear_system_solvers/julia.jl:21-113
 * whose arithmetic lives in a third-party dependency that is NOT under /root/reference:
 * SuiteSparse CHOLMOD reached through Julia's stdlib (`ldlt(Symmetric(A,:L))`, julia.jl:52;
 * `cholesky(Symmetric(A,:L))`, julia.jl:34; `F \ b`, julia.jl:101,110).  Julia 1.7.1 bundles
 * SuiteSparse 5.10.1 / CHOLMOD 3.0.14 (version from Julia's release notes, not from the repo).
 * For `ldlt` CHOLMOD runs a SIMPLICIAL, up-looking LDL^T WITHOUT numerical pivoting: row k of L
 * is obtained by a sparse triangular solve along the elimination-tree reach of the pattern of
 * A(1:k,k); negative pivots are accepted and an exactly-zero pivot aborts (ZeroPivotException,
 * which julia.jl:61-63 maps to inertia flag 0).  That published algorithm (T. A. Davis,
 * "Algorithm 849: a concise sparse Cholesky factorization package", ACM TOMS 31, 2005, and
 * "Direct Methods for Sparse Linear Systems", SIAM 2006, ch. 4) is restated below in our own
 * code: elimination tree + column counts, up-looking numeric phase, L / D / L^T solves,
 * permutation handling, and the inertia rule of julia.jl:70-90 +
 * linear_system_solvers.jl:48-91.
 *
 * PARITY STATUS: the oracle is pinned against (a) the matrices and acceptance rules of the
 * reference's own unit tests (test/linear_system_solvers.jl:94-116, test/kkt_system_solvers.jl
 * :91-181) and (b) dense LAPACK answers (numpy) -- see tests/test_oracle.py.  It is UNPINNED
 * against CHOLMOD's internal choices (its AMD permutation and the resulting D values): neither
 * Julia nor SuiteSparse can run in this environment and the reference stores no golden factors.
 * The permutation is therefore an INPUT of the oracle (natural order when none is given); parity
 * tests feed it the permutation the product computed, so "same pivot order" holds by
 * construction and the comparison isolates the numeric phase.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct oracle_factor {
  int64_t n;
  int64_t *perm, *iperm;   /* perm[new] = old */
  int64_t *parent;         /* elimination tree */
  int64_t *Lp, *Li;        /* strictly-lower L by columns, rows in the order the up-looking sweep appends them */
  double *Lx, *D;
  int64_t *Cp, *Ci;        /* permuted matrix, upper triangle by columns (rows <= col) */
  double *Cx;
  int64_t *cmap;           /* for each input entry: slot in C, or -1 when ignored (row < col) */
  int64_t nnz_in;
  int64_t lnz;             /* nnz(L) strictly lower */
  double flops;            /* sum_j c_j^2 with c_j = column count incl. diagonal */
  int64_t npiv_done;       /* pivots completed by the last numeric phase */
  /* work */
  double *Y;
  int64_t *Pattern, *Flag, *Lnz;
} oracle_factor;

void oracle_free(oracle_factor *F) {
  if (!F) return;
  free(F->perm); free(F->iperm); free(F->parent); free(F->Lp); free(F->Li); free(F->Lx); free(F->D);
  free(F->Cp); free(F->Ci); free(F->Cx); free(F->cmap); free(F->Y); free(F->Pattern); free(F->Flag); free(F->Lnz);
  free(F);
}

/* julia.jl:34,52: Symmetric(A,:L) -- only entries with row >= col are read.
 * perm may be NULL (natural order).  Returns NULL on invalid input. */
oracle_factor *oracle_analyze(int64_t n, const int64_t *colptr, const int64_t *rowval, int index_base,
                              const int64_t *perm) {
  oracle_factor *F = (oracle_factor *)calloc(1, sizeof(oracle_factor));
  if (!F) return NULL;
  const int64_t base = index_base;
  const int64_t nnz = colptr[n] - base;
  F->n = n;
  F->nnz_in = nnz;
  F->perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->iperm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->parent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->Lp = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t));
  F->Cp = (int64_t *)calloc((size_t)(n + 2), sizeof(int64_t));
  F->cmap = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
  F->D = (double *)calloc((size_t)(n + 1), sizeof(double));
  F->Y = (double *)calloc((size_t)(n + 1), sizeof(double));
  F->Pattern = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->Flag = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  F->Lnz = (int64_t *)calloc((size_t)(n + 1), sizeof(int64_t));
  for (int64_t k = 0; k < n; ++k) { F->perm[k] = perm ? perm[k] : k; }
  for (int64_t k = 0; k < n; ++k) F->iperm[k] = -1;
  for (int64_t k = 0; k < n; ++k) {
    int64_t o = F->perm[k];
    if (o < 0 || o >= n || F->iperm[o] != -1) { oracle_free(F); return NULL; }
    F->iperm[o] = k;
  }
  /* C = upper triangle of P A P', by columns: entry (i,j), i >= j  ->  column max(pi,pj), row min(pi,pj) */
  for (int64_t j = 0; j < n; ++j)
    for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
      int64_t i = rowval[p] - base;
      if (i < 0 || i >= n) { oracle_free(F); return NULL; }
      if (i < j) continue;
      int64_t a = F->iperm[i], b = F->iperm[j];
      F->Cp[(a > b ? a : b) + 1]++;
    }
  for (int64_t j = 0; j < n; ++j) F->Cp[j + 1] += F->Cp[j];
  const int64_t cnz = F->Cp[n];
  F->Ci = (int64_t *)malloc(sizeof(int64_t) * (size_t)(cnz + 1));
  F->Cx = (double *)calloc((size_t)(cnz + 1), sizeof(double));
  int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  memcpy(fill, F->Cp, sizeof(int64_t) * (size_t)n);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
      int64_t i = rowval[p] - base;
      if (i < j) { F->cmap[p] = -1; continue; }
      int64_t a = F->iperm[i], b = F->iperm[j];
      int64_t col = a > b ? a : b, row = a > b ? b : a;
      int64_t q = fill[col]++;
      F->Ci[q] = row;
      F->cmap[p] = q;
    }
  free(fill);
  /* elimination tree and column counts (row-subtree traversal) */
  for (int64_t k = 0; k < n; ++k) {
    F->parent[k] = -1;
    F->Flag[k] = k;
    F->Lnz[k] = 0;
    for (int64_t p = F->Cp[k]; p < F->Cp[k + 1]; ++p) {
      int64_t i = F->Ci[p];
      for (; i < k && F->Flag[i] != k; i = F->parent[i]) {
        if (F->parent[i] == -1) F->parent[i] = k;
        F->Lnz[i]++;
        F->Flag[i] = k;
      }
    }
  }
  F->Lp[0] = 0;
  F->flops = 0;
  for (int64_t k = 0; k < n; ++k) {
    F->Lp[k + 1] = F->Lp[k] + F->Lnz[k];
    double c = (double)(F->Lnz[k] + 1);
    F->flops += c * c;
  }
  F->lnz = F->Lp[n];
  F->Li = (int64_t *)malloc(sizeof(int64_t) * (size_t)(F->lnz + 1));
  F->Lx = (double *)malloc(sizeof(double) * (size_t)(F->lnz + 1));
  return F;
}

/* Up-looking LDL^T without pivoting.  Returns 1 when all n pivots were computed, 0 when an
 * exactly-zero pivot stopped the factorisation (ZeroPivotException in the reference's ldlt). */
int oracle_numeric(oracle_factor *F, const double *nzval) {
  const int64_t n = F->n;
  for (int64_t q = 0; q < F->Cp[n]; ++q) F->Cx[q] = 0.0;
  for (int64_t p = 0; p < F->nnz_in; ++p)
    if (F->cmap[p] >= 0) F->Cx[F->cmap[p]] += nzval[p];
  double *Y = F->Y;
  int64_t *Pattern = F->Pattern, *Flag = F->Flag, *Lnz = F->Lnz;
  F->npiv_done = 0;
  for (int64_t k = 0; k < n; ++k) {
    Y[k] = 0.0;
    int64_t top = n;
    Flag[k] = k;
    Lnz[k] = 0;
    for (int64_t p = F->Cp[k]; p < F->Cp[k + 1]; ++p) {
      int64_t i = F->Ci[p];
      if (i > k) continue;
      Y[i] += F->Cx[p];
      int64_t len = 0;
      for (; Flag[i] != k; i = F->parent[i]) { Pattern[len++] = i; Flag[i] = k; }
      while (len > 0) Pattern[--top] = Pattern[--len];
    }
    double dk = Y[k];
    Y[k] = 0.0;
    for (; top < n; ++top) {
      int64_t i = Pattern[top];
      double yi = Y[i];
      Y[i] = 0.0;
      int64_t p2 = F->Lp[i] + Lnz[i];
      for (int64_t p = F->Lp[i]; p < p2; ++p) Y[F->Li[p]] -= F->Lx[p] * yi;
      double lki = yi / F->D[i];
      dk -= lki * yi;
      F->Li[p2] = k;
      F->Lx[p2] = lki;
      Lnz[i]++;
    }
    F->D[k] = dk;
    if (dk == 0.0) { F->npiv_done = k; return 0; }
  }
  F->npiv_done = n;
  return 1;
}

/* julia.jl:72-78 */
void oracle_inertia(const oracle_factor *F, double tol, int64_t *pos, int64_t *neg, int64_t *zero, int64_t *nonfinite) {
  int64_t a = 0, b = 0, c = 0, d = 0;
  for (int64_t k = 0; k < F->n; ++k) {
    double x = F->D[k];
    if (isnan(x) || isinf(x)) ++d;
    else if (x > tol) ++a;
    else if (x < -tol) ++b;
    else ++c;
  }
  *pos = a; *neg = b; *zero = c; *nonfinite = d;
}

/* ls_factor! (julia.jl:21-97).  sym_kind 0 = :definite, 1 = :symmetric.  Returns 1 / 0, -1 on misuse. */
int oracle_ls_factor(oracle_factor *F, const double *nzval, int64_t n, int64_t m, int sym_kind) {
  if (n + m != F->n) return -1;
  if (sym_kind == 0) {
    if (m != 0) return -1;                      /* @assert(m == 0), julia.jl:30 */
    int done = oracle_numeric(F, nzval);
    /* cholesky succeeds iff every pivot is > 0 (PosDefException otherwise, julia.jl:39-41) */
    if (!done) return 0;
    for (int64_t k = 0; k < F->n; ++k) if (!(F->D[k] > 0.0)) return 0;
    return 1;
  }
  if (!oracle_numeric(F, nzval)) return 0;      /* ZeroPivotException -> 0, julia.jl:61-63 */
  int64_t pos, neg, zero, bad;
  oracle_inertia(F, 1e-20, &pos, &neg, &zero, &bad);
  if (bad > 0) return 0;                        /* julia.jl:77-89 */
  if (pos + neg + zero != n + m) return -1;     /* linear_system_solvers.jl:62-69 */
  return (pos == n && neg == m) ? 1 : 0;        /* linear_system_solvers.jl:73-74 */
}

/* sol = P' L^-T D^-1 L^-1 P rhs   (F \ b, julia.jl:101,110) */
void oracle_solve(const oracle_factor *F, const double *rhs, double *sol) {
  const int64_t n = F->n;
  double *x = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  for (int64_t k = 0; k < n; ++k) x[k] = rhs[F->perm[k]];
  for (int64_t j = 0; j < n; ++j) {
    double xj = x[j];
    for (int64_t p = F->Lp[j]; p < F->Lp[j + 1]; ++p) x[F->Li[p]] -= F->Lx[p] * xj;
  }
  for (int64_t j = 0; j < n; ++j) x[j] /= F->D[j];
  for (int64_t j = n - 1; j >= 0; --j) {
    double xj = x[j];
    for (int64_t p = F->Lp[j]; p < F->Lp[j + 1]; ++p) xj -= F->Lx[p] * x[F->Li[p]];
    x[j] = xj;
  }
  for (int64_t k = 0; k < n; ++k) sol[F->perm[k]] = x[k];
  free(x);
}

/* accessors for ctypes */
int64_t oracle_n(const oracle_factor *F) { return F->n; }
int64_t oracle_lnz(const oracle_factor *F) { return F->lnz; }
double oracle_flops(const oracle_factor *F) { return F->flops; }
int64_t oracle_npiv_done(const oracle_factor *F) { return F->npiv_done; }
void oracle_get_D(const oracle_factor *F, double *out) { memcpy(out, F->D, sizeof(double) * (size_t)F->n); }
void oracle_get_parent(const oracle_factor *F, int64_t *out) { memcpy(out, F->parent, sizeof(int64_t) * (size_t)F->n); }
void oracle_get_colcounts(const oracle_factor *F, int64_t *out) {
  for (int64_t k = 0; k < F->n; ++k) out[k] = F->Lp[k + 1] - F->Lp[k] + 1;
}
void oracle_get_L(const oracle_factor *F, int64_t *Lp, int64_t *Li, double *Lx) {
  memcpy(Lp, F->Lp, sizeof(int64_t) * (size_t)(F->n + 1));
  memcpy(Li, F->Li, sizeof(int64_t) * (size_t)F->lnz);
  memcpy(Lx, F->Lx, sizeof(double) * (size_t)F->lnz);
}
