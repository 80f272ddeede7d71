// Which (XCC, SE, SH, CU) does bit b of a hipExtStreamCreateWithCUMask mask select?  One single-bit stream per bit of XCD 0 .. 1, a tiny kernel on each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* o) { int x, h; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h)); if (threadIdx.x == 0) { o[0] = x & 15; o[1] = h; } }
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  int* d; hipMalloc(&d, 8);
  for (int b = 0; b < ncu; ++b) {
    if ((b & 7) > 1) continue;      // XCD 0 and 1 only
    std::vector<uint32_t> m((ncu + 31) / 32, 0u); m[b >> 5] |= 1u << (b & 31);
    hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, m.size(), m.data()) != hipSuccess) { printf("bit %d: stream failed\n", b); continue; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s);
    int h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("bit %3d (cu index %2d of xcd slot %d): xcc %d se %d sh %d cu %2d\n", b, b >> 3, b & 7, h[0], (h[1] >> 13) & 7, (h[1] >> 12) & 1, (h[1] >> 8) & 15);
    hipStreamDestroy(s);
  }
  return 0;
}
