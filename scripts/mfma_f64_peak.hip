// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (operands in registers).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_f64_peak.hip -o /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_peak(double* out, const double* in, int iters) {
  d4_t acc[NACC];
  const double a0 = in[threadIdx.x], b0 = in[threadIdx.x + 256];
  double a = a0, b = b0;
  for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, const char* label) {
  const int iters = 4000;
  const int nblk = 256 * blocks_per_cu;
  double *out, *in;
  hipMalloc(&out, nblk * 256 * 8);
  hipMalloc(&in, 512 * 8);
  std::vector<double> h(512);
  for (int i = 0; i < 512; ++i) h[i] = 1.0 + 1e-3 * (i % 97) - 0.04;   // non-trivial data (DVFS)
  hipMemcpy(in, h.data(), 512 * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(nblk), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)nblk * 4 /*waves*/ * iters * NACC * 2048.0;
    if (rep == 2) printf("%s: NACC=%d blocks/CU=%d  %.3f ms  %.2f TFLOP/s\n", label, NACC, blocks_per_cu, ms, flops / ms / 1e9);
  }
  hipFree(out); hipFree(in);
}

int main() {
  run<4>(1, "1 wave/SIMD");
  run<16>(1, "1 wave/SIMD");
  run<4>(2, "2 waves/SIMD");
  run<16>(2, "2 waves/SIMD");
  run<16>(4, "4 waves/SIMD");
  return 0;
}
