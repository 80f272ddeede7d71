/* A caller of the C ABI that is neither Python nor Julia: plain C, compiled with gcc against
 * include/okkt.h and linked to libonephase_kkt.so.  It re-enacts the reference's test_linear_solvers on the device
 * (/root/reference/test/linear_system_solvers.jl:58-116): the 10 x 10 identity and the identity with A[10,1] = A[9,2] = 0.1, as
 * 1-based Int64 CSC like Julia's SparseMatrixCSC, through okkt_create -> okkt_analyze -> okkt_factor -> okkt_solve for :symmetric and
 * :definite, on the lower-only matrix and on A + A' (the upper entries must be ignored).  The right-hand side comes from the command
 * line (ten numbers), every solution is printed (%.17g) for the calling test to compare with the golden answers; the program checks
 * the reference's own rules itself (inertia flag 1, |x_sym - x_chol| < 1e-9, upper-triangle invariance) and returns 1 on a violation. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "okkt.h"

#define N 10
static int fail(const char* what, okkt_handle h, int rc) {
  fprintf(stderr, "abi_driver: %s failed (%d): %s\n", what, rc, h ? okkt_last_error(h) : "");
  return 1;
}

/* CSC of I (+ the two off-diagonal entries when offd), lower only or with the mirrored upper entries; 1-based */
static int build(int offd, int upper, int64_t* colptr, int64_t* rowval, double* nzval) {
  int nnz = 0;
  for (int j = 1; j <= N; ++j) {
    colptr[j - 1] = nnz + 1;
    if (offd && upper && j == 10) { rowval[nnz] = 1; nzval[nnz++] = 0.1; }     /* A'[1,10] */
    if (offd && upper && j == 9) { rowval[nnz] = 2; nzval[nnz++] = 0.1; }      /* A'[2,9]  */
    rowval[nnz] = j; nzval[nnz++] = 1.0;
    if (offd && j == 1) { rowval[nnz] = 10; nzval[nnz++] = 0.1; }              /* A[10,1] */
    if (offd && j == 2) { rowval[nnz] = 9; nzval[nnz++] = 0.1; }               /* A[9,2]  */
  }
  colptr[N] = nnz + 1;
  return nnz;
}

static int run(int offd, int upper, int kind, const double* b, double* x) {
  int64_t colptr[N + 1], rowval[N + 4];
  double nzval[N + 4];
  build(offd, upper, colptr, rowval, nzval);
  okkt_handle h = NULL;
  okkt_opts o;
  okkt_default_opts(&o);
  int rc = okkt_create(&h, &o);
  if (rc != OKKT_OK) return fail("okkt_create (no HIP device? there is no CPU fallback)", NULL, rc);
  if ((rc = okkt_analyze(h, N, colptr, rowval, 1)) != OKKT_OK) return fail("okkt_analyze", h, rc);
  okkt_inertia in;
  rc = okkt_factor(h, nzval, N, 0, kind, &in);                      /* n = 10, m = 0, as the reference's test */
  if (rc != 1) return fail("okkt_factor: inertia flag", h, rc);
  if (in.pos != N || in.neg != 0 || in.zero != 0 || in.nonfinite != 0) return fail("inertia counts", h, (int)in.pos);
  if ((rc = okkt_solve(h, b, x, 1)) != OKKT_OK) return fail("okkt_solve", h, rc);
  okkt_destroy(h);
  return 0;
}

int main(int argc, char** argv) {
  double b[N], x[2][2][2][N];      /* [matrix][upper filled][kind] */
  if (argc != N + 1) { fprintf(stderr, "usage: abi_driver b1 ... b10\n"); return 2; }
  for (int i = 0; i < N; ++i) b[i] = atof(argv[i + 1]);
  for (int offd = 0; offd < 2; ++offd)
    for (int upper = 0; upper < 2; ++upper)
      for (int kind = 0; kind < 2; ++kind) {      /* 0 = OKKT_SYM_DEFINITE (cholesky), 1 = OKKT_SYM_SYMMETRIC (ldlt) */
        if (run(offd, upper, kind == 0 ? OKKT_SYM_DEFINITE : OKKT_SYM_SYMMETRIC, b, x[offd][upper][kind])) return 1;
        printf("x %d %d %d", offd, upper, kind);
        for (int i = 0; i < N; ++i) printf(" %.17g", x[offd][upper][kind][i]);
        printf("\n");
      }
  for (int offd = 0; offd < 2; ++offd) {
    double d_kind = 0, d_upper = 0;
    for (int i = 0; i < N; ++i) {
      d_kind = fmax(d_kind, fabs(x[offd][0][0][i] - x[offd][0][1][i]));          /* test/linear_system_solvers.jl:62,67 */
      for (int kind = 0; kind < 2; ++kind) d_upper = fmax(d_upper, fabs(x[offd][0][kind][i] - x[offd][1][kind][i]));   /* :74-84 */
    }
    if (d_kind >= 1e-9 || d_upper >= 1e-9) { fprintf(stderr, "abi_driver: matrix %d: sym vs chol %.3g, upper-triangle %.3g\n", offd, d_kind, d_upper); return 1; }
  }
  printf("ok\n");
  return 0;
}
