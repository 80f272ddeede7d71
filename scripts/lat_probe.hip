// Latency table for single-wave, latency-bound code on gfx950 (one workgroup on an otherwise idle GPU):
// dependent VALU chains, v_rcp_f64, LDS round trips, s_barrier, MFMA chains.  Times from wall_clock64 (100 MHz).
// build: hipcc --offload-arch=gfx950 -O2 scripts/lat_probe.hip -o scripts/_bin/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int N = 20000;

__global__ void k_lat(double* out, long long* t, int nwaves_active) {
  __shared__ double lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  long long t0, t1;
  int slot = 0;
  // 1: dependent f32 add chain
  { float a = (float)tid; __syncthreads(); t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) a = a * 1.0001f + 0.5f;
    t1 = wall_clock64(); out[tid] = a; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 2: dependent f64 fma chain
  { double a = (double)tid; __syncthreads(); t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) a = __builtin_fma(a, 1.0000001, 0.5);
    t1 = wall_clock64(); out[tid] += a; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 3: 4 independent f64 fma chains (issue rate)
  { double a = tid, b = tid + 1, c = tid + 2, d = tid + 3; __syncthreads(); t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { a = __builtin_fma(a, 1.0000001, 0.5); b = __builtin_fma(b, 1.0000001, 0.5); c = __builtin_fma(c, 1.0000001, 0.5); d = __builtin_fma(d, 1.0000001, 0.5); }
    t1 = wall_clock64(); out[tid] += a + b + c + d; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 4: dependent v_rcp_f64 chain
  { double a = 1.5 + tid; __syncthreads(); t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) a = __builtin_amdgcn_rcp(a) + 1.0;
    t1 = wall_clock64(); out[tid] += a; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 5: LDS write -> read round trip (dependent)
  { double a = tid; __syncthreads(); t0 = wall_clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) { lds[tid] = a; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a = lds[tid ^ 1] + 1.0; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    t1 = wall_clock64(); out[tid] += a; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 6: s_barrier (all waves of the workgroup)
  { __syncthreads(); t0 = wall_clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) { __builtin_amdgcn_s_barrier(); }
    t1 = wall_clock64(); if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 7: dependent mfma 16x16x4 f64 chain
  { d4_t acc = {0, 0, 0, 0}; double a = 1.0 + lane * 1e-9, b = 1.0; __syncthreads(); t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    t1 = wall_clock64(); out[tid] += acc[0]; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 8: dependent mfma 4x4x4 f64 chain
  { double acc = 0, a = 1.0 + lane * 1e-9, b = 1.0; __syncthreads(); t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    t1 = wall_clock64(); out[tid] += acc; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 9: readlane -> valu chain
  { double a = tid; __syncthreads(); t0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(a), 3); a = a + (double)lo; }
    t1 = wall_clock64(); out[tid] += a; if (tid == 0) t[slot] = t1 - t0; ++slot; }
  // 10: LDS broadcast read chain (address depends on previous value)
  { int idx = tid & 7; lds[tid] = (double)((tid * 7) & 63); __syncthreads(); t0 = wall_clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) { idx = (int)lds[idx]; }
    t1 = wall_clock64(); out[tid] += idx; if (tid == 0) t[slot] = t1 - t0; ++slot; }
}

int main() {
  double* out; long long* t;
  CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&t, 64 * 8));
  const char* names[] = {"dep f32 fma", "dep f64 fma", "4 indep f64 fma (per 4)", "dep v_rcp_f64 + add", "LDS write->read round trip", "s_barrier",
                         "dep mfma f64 16x16x4", "dep mfma f64 4x4x4", "readlane + cvt + add", "dep LDS read (+cvt)"};
  for (int threads : {64, 256}) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_lat, dim3(1), dim3(threads), 0, 0, out, t, 0); CK(hipDeviceSynchronize()); }
    long long h[16]; CK(hipMemcpy(h, t, 10 * 8, hipMemcpyDeviceToHost));
    printf("threads = %d\n", threads);
    for (int i = 0; i < 10; ++i) printf("  %-28s %8.2f ns per iteration\n", names[i], h[i] * 10.0 / N);
  }
  return 0;
}
