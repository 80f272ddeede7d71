// f64 MFMA issue-rate experiments: accumulators in VGPRs vs AGPRs, dependency distance, waves/SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC, bool AGPR>
__global__ __launch_bounds__(256) void k_peak(double* out, const double* in, int iters) {
  d4_t acc[NACC];
  double a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (AGPR) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 4x4x4 (4 blocks) form: 512 flops per instruction
template <int NACC>
__global__ __launch_bounds__(256) void k_peak4(double* out, const double* in, int iters) {
  double acc[NACC];
  double a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// plain vector FMA f64 for comparison
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, const double* in, int iters) {
  double acc[NACC];
  double a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.1 * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
void timeit(K launch, double flops, const char* label) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) best = ms < best ? ms : best;
  }
  printf("%-44s %.3f ms  %.2f TFLOP/s\n", label, best, flops / best / 1e9);
}

int main() {
  double *out, *in;
  hipMalloc(&out, 4096 * 256 * 8); hipMalloc(&in, 512 * 8);
  std::vector<double> h(512);
  for (int i = 0; i < 512; ++i) h[i] = 1.0 + 1e-3 * (i % 97) - 0.04;
  hipMemcpy(in, h.data(), 512 * 8, hipMemcpyHostToDevice);
  const int it = 3000;
#define RUN(KERN, NACC, BPC, FL, LABEL) timeit([&] { hipLaunchKernelGGL(KERN, dim3(256 * BPC), dim3(256), 0, 0, out, in, it); }, (double)256 * BPC * 4 * it * NACC * FL, LABEL)
  RUN((k_peak<1, false>), 1, 1, 2048.0, "16x16x4 VGPR acc, NACC=1, 1 w/SIMD");
  RUN((k_peak<2, false>), 2, 1, 2048.0, "16x16x4 VGPR acc, NACC=2, 1 w/SIMD");
  RUN((k_peak<8, false>), 8, 1, 2048.0, "16x16x4 VGPR acc, NACC=8, 1 w/SIMD");
  RUN((k_peak<1, false>), 1, 2, 2048.0, "16x16x4 VGPR acc, NACC=1, 2 w/SIMD");
  RUN((k_peak<2, false>), 2, 2, 2048.0, "16x16x4 VGPR acc, NACC=2, 2 w/SIMD");
  RUN((k_peak<1, false>), 1, 4, 2048.0, "16x16x4 VGPR acc, NACC=1, 4 w/SIMD");
  RUN((k_peak<2, false>), 2, 4, 2048.0, "16x16x4 VGPR acc, NACC=2, 4 w/SIMD");
  RUN((k_peak<4, false>), 4, 4, 2048.0, "16x16x4 VGPR acc, NACC=4, 4 w/SIMD");
  RUN((k_peak<1, false>), 1, 8, 2048.0, "16x16x4 VGPR acc, NACC=1, 8 w/SIMD");
  RUN((k_peak<16, true>), 16, 1, 2048.0, "16x16x4 AGPR acc, NACC=16, 1 w/SIMD");
  RUN((k_peak<16, true>), 16, 2, 2048.0, "16x16x4 AGPR acc, NACC=16, 2 w/SIMD");
  RUN((k_peak<4, true>), 4, 2, 2048.0, "16x16x4 AGPR acc, NACC=4, 2 w/SIMD");
  RUN((k_peak4<8>), 8, 2, 512.0, "4x4x4 NACC=8, 2 w/SIMD");
  RUN((k_peak4<16>), 16, 4, 512.0, "4x4x4 NACC=16, 4 w/SIMD");
  RUN((k_fma<16>), 16, 2, 128.0, "v_fma_f64 NACC=16, 2 w/SIMD");
  RUN((k_fma<16>), 16, 4, 128.0, "v_fma_f64 NACC=16, 4 w/SIMD");
  return 0;
}
