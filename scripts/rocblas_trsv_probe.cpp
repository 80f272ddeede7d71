// How fast is rocBLAS's dtrsv / dtrsm / dgemv on the shapes of the root front?  (decision aid for the solve phase)
// build: hipcc --offload-arch=gfx950 -O2 scripts/rocblas_trsv_probe.cpp -o scripts/_bin/rocblas_trsv_probe -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { auto e = (x); if ((int)e != 0) { printf("error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)
int main() {
  const int n = 16641;
  rocblas_handle h; CK(rocblas_create_handle(&h));
  hipStream_t st; CK(hipStreamCreate(&st)); CK(rocblas_set_stream(h, st));
  double *A, *x, *X, *B;
  CK(hipMalloc(&A, (size_t)n * n * 8)); CK(hipMalloc(&x, (size_t)n * 8));
  std::vector<double> hx(n, 1.0);
  // A = I + small strictly lower entries
  std::vector<double> col(n);
  CK(hipMemset(A, 0, (size_t)n * n * 8));
  for (int j = 0; j < n; j += 97) { for (int i = 0; i < n; ++i) col[i] = i > j ? 1e-4 : (i == j ? 2.0 : 0.0); CK(hipMemcpy(A + (size_t)j * n, col.data(), n * 8, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int tr = 0; tr < 2; ++tr) {
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0, st));
      CK(rocblas_dtrsv(h, rocblas_fill_lower, tr ? rocblas_operation_transpose : rocblas_operation_none, rocblas_diagonal_unit, n, A, n, x, 1));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("dtrsv %s n=%d: %.3f ms\n", tr ? "T" : "N", n, ms);
    }
  }
  // explicit inverse of 1024-blocks by dtrsm on identity (batched over the diagonal blocks)
  for (int sb : {512, 1024}) {
    const int nb = n / sb;
    CK(hipMalloc(&X, (size_t)nb * sb * sb * 8));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(X, 0, (size_t)nb * sb * sb * 8));
      std::vector<double> eye((size_t)sb * sb, 0.0); for (int i = 0; i < sb; ++i) eye[(size_t)i * sb + i] = 1.0;
      for (int b = 0; b < nb; ++b) CK(hipMemcpy(X + (size_t)b * sb * sb, eye.data(), (size_t)sb * sb * 8, hipMemcpyHostToDevice));
      const double one = 1.0;
      CK(hipEventRecord(e0, st));
      CK(rocblas_dtrsm_strided_batched(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_unit, sb, sb, &one,
                                       A, n, (rocblas_stride)sb * (n + 1), X, sb, (rocblas_stride)sb * sb, nb));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("dtrsm inverse of %d blocks of %d: %.3f ms\n", nb, sb, ms);
    }
    CK(hipFree(X));
  }
  // gemv on a tall panel (rows below x 1024 columns)
  CK(hipMalloc(&B, (size_t)n * 8));
  for (int rep = 0; rep < 3; ++rep) {
    const double m1 = -1.0, one = 1.0;
    CK(hipEventRecord(e0, st));
    CK(rocblas_dgemv(h, rocblas_operation_none, n - 1024, 1024, &m1, A + 1024, n, x, 1, &one, B, 1));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dgemv (n-1024) x 1024: %.3f ms\n", ms);
  }
  return 0;
}
