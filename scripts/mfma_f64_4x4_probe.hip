// Determines the lane layout of v_mfma_f64_4x4x4_4b_f64 (A, B, D, cbsz/abid broadcast) and its rate
// with many accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

template <int CBSZ, int ABID>
__global__ void k_one(const double* a, const double* b, double* d) {
  double acc = 0.0;
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], acc, CBSZ, ABID, 0);
  d[threadIdx.x] = acc;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_rate(double* out, const double* in, int iters) {
  double acc[NACC];
  double a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], b0 = in[threadIdx.x + 256], b1 = in[threadIdx.x + 300];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; i += 8) {
      acc[i + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc[i + 0], 2, 0, 0);
      acc[i + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc[i + 1], 2, 1, 0);
      acc[i + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc[i + 2], 2, 2, 0);
      acc[i + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc[i + 3], 2, 3, 0);
      acc[i + 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc[i + 4], 2, 0, 0);
      acc[i + 5] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc[i + 5], 2, 1, 0);
      acc[i + 6] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc[i + 6], 2, 2, 0);
      acc[i + 7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc[i + 7], 2, 3, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  std::vector<double> a(64), b(64), d(64);
  for (int l = 0; l < 64; ++l) { a[l] = 1.0 + l * 0.37 + (l % 7) * 0.011; b[l] = 2.0 - l * 0.21 + (l % 5) * 0.013; }
  double *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
  auto check = [&](const char* name, int cbsz, int abid) {
    hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    // candidate maps: A lane -> (i,k): mode 0: i = l%4, k = (l/4)%4 ; mode 1: swapped.  same for B (k,j) and D (i,j)
    for (int ma = 0; ma < 2; ++ma) for (int mb = 0; mb < 2; ++mb) for (int md = 0; md < 2; ++md) {
      double err = 0;
      for (int l = 0; l < 64; ++l) {
        int blk = l / 16, x = l % 4, y = (l / 4) % 4;
        int i = md == 0 ? x : y, j = md == 0 ? y : x;
        int ablk = cbsz == 2 ? abid : blk;
        double s = 0;
        for (int k = 0; k < 4; ++k) {
          int la = ma == 0 ? (ablk * 16 + i + 4 * k) : (ablk * 16 + k + 4 * i);   // lane holding A[i][k]
          int lb = mb == 0 ? (blk * 16 + k + 4 * j) : (blk * 16 + j + 4 * k);     // lane holding B[k][j]
          s += a[la] * b[lb];
        }
        err = fmax(err, fabs(s - d[l]));
      }
      if (err < 1e-9) printf("%s: MATCH A mode %d (0: i=l%%4,k=l/4 | 1: k=l%%4,i=l/4), B mode %d (0: k=l%%4,j=l/4 | 1: j=l%%4,k=l/4), D mode %d (0: i=l%%4,j=l/4 | 1: j=l%%4,i=l/4)\n", name, ma, mb, md);
    }
  };
  hipLaunchKernelGGL((k_one<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz0", 0, 0);
  hipLaunchKernelGGL((k_one<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz2 abid0", 2, 0);
  hipLaunchKernelGGL((k_one<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz2 abid3", 2, 3);
  hipLaunchKernelGGL((k_one<1, 1>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize();
  hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost); printf("cbsz1 abid1 d[0..3]= %g %g %g %g d[16]=%g d[32]=%g d[48]=%g\n", d[0], d[1], d[2], d[3], d[16], d[32], d[48]);
  // rate with many accumulators
  double *out, *in; hipMalloc(&out, 2048 * 256 * 8); hipMalloc(&in, 1024 * 8);
  std::vector<double> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 1.0 + 1e-3 * (i % 97) - 0.04;
  hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto rate = [&](auto kern, int nacc, int bpc, const char* label) {
    const int it = 1500; float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256 * bpc), dim3(256), 0, 0, out, in, it); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep) best = fminf(best, ms); }
    printf("%-40s %.3f ms %.2f TFLOP/s\n", label, best, (double)256 * bpc * 4 * it * nacc * 512.0 / best / 1e9);
  };
  rate(k_rate<64>, 64, 1, "4x4x4 cbsz2 NACC=64, 1 w/SIMD");
  rate(k_rate<64>, 64, 2, "4x4x4 cbsz2 NACC=64, 2 w/SIMD");
  rate(k_rate<32>, 32, 2, "4x4x4 cbsz2 NACC=32, 2 w/SIMD");
  rate(k_rate<16>, 16, 1, "4x4x4 cbsz2 NACC=16, 1 w/SIMD");
  return 0;
}
