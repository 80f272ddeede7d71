// Which SIMD does each wave of a 512- and a 1024-thread workgroup land on?  (HW_REG_HW_ID bits 5:4 = SIMD, 3:0 = wave slot.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 16 * 4);
  for (int threads : {512, 1024}) {
    hipMemset(d, 0xff, 4 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(threads), 0, 0, d);
    unsigned h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
      printf("threads %d wg %d: simd of waves:", threads, b);
      for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   cu %u\n", (h[b * 16] >> 8) & 15);
    }
  }
  return 0;
}
