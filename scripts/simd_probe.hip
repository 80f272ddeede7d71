// which SIMD does wave w of a 512-thread workgroup run on?  (HW_ID bits 5:4 = SIMD id, 11:8 = CU id on gfx9-family)
// build: hipcc -O2 --offload-arch=gfx950 scripts/simd_probe.hip -o scripts/_bin/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 8 * 16);
  hipLaunchKernelGGL(k, dim3(16), dim3(512), 100 * 1024, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 16; ++b) {
    printf("wg %2d cu %2u se %u:", b, (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
    for (int w = 0; w < 8; ++w) printf(" w%d->simd%u", w, (h[b * 8 + w] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
