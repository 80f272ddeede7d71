#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void k_dec(double* d) {
  const int la = blockIdx.x / 64, lb = blockIdx.x % 64;
  double a = threadIdx.x == la ? 1.0 : 0.0, b = threadIdx.x == lb ? 1.0 : 0.0;
  double acc = 0.0;
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, CBSZ, ABID, 0);
  d[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
  double* dd; hipMalloc(&dd, 4096 * 64 * 8);
  std::vector<double> d(4096 * 64);
  hipLaunchKernelGGL((k_dec<0, 0>), dim3(4096), dim3(64), 0, 0, dd); hipDeviceSynchronize();
  hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost);
  // for each (la, lb): list output lanes that are 1
  int shown = 0;
  printf("cbsz=0: (la,lb) -> output lanes\n");
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
    int cnt = 0, first = -1;
    for (int l = 0; l < 64; ++l) if (d[(la * 64 + lb) * 64 + l] != 0.0) { ++cnt; if (first < 0) first = l; }
    if (cnt && (la < 20 || la == 33) && shown < 400) { printf("(%d,%d)->%d%s ", la, lb, first, cnt > 1 ? "+" : ""); ++shown; if (shown % 8 == 0) printf("\n"); }
  }
  printf("\n");
  hipLaunchKernelGGL((k_dec<2, 1>), dim3(4096), dim3(64), 0, 0, dd); hipDeviceSynchronize();
  hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost);
  printf("cbsz=2 abid=1: pairs producing output, la in {16,17,21}\n");
  for (int la : {0, 16, 17, 21}) for (int lb = 0; lb < 64; ++lb) {
    int cnt = 0, first = -1;
    for (int l = 0; l < 64; ++l) if (d[(la * 64 + lb) * 64 + l] != 0.0) { ++cnt; if (first < 0) first = l; }
    if (cnt) printf("(%d,%d)->%d%s ", la, lb, first, cnt > 1 ? "+" : "");
  }
  printf("\n");
  return 0;
}
