// Residency census: how many 512-thread workgroups of a 128-VGPR kernel really share a CU on a CU-masked stream, by LDS size and scratch use.
// build: hipcc -O3 --offload-arch=gfx950 scripts/occ_probe.hip -o scripts/_bin/occ_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
template <int SCR, int NA = 24>
__global__ __launch_bounds__(512, 4) void k(int* out, int spin_us, double* sink) {
  extern __shared__ double sm[];
  double acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double loc[SCR > 0 ? SCR : 1];
  if (SCR > 0) { for (int i = 0; i < SCR; ++i) loc[i] = i * 0.5 + threadIdx.x; }
  int xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = xcc & 15; out[blockIdx.x * 4 + 1] = hw; out[blockIdx.x * 4 + 3] = (int)(t0 & 0x7fffffff); }
  sm[threadIdx.x] = acc[3];
  int it = 0;
  while (wall_clock64() - t0 < (long long)spin_us * 100) {
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = acc[i] * 1.0000001 + sm[(threadIdx.x + i) & 511];
    if (SCR > 0) loc[(it++) % SCR] += acc[5];
    __builtin_amdgcn_s_sleep(4);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) s += acc[i];
  if (SCR > 0) for (int i = 0; i < SCR; ++i) s += loc[i];
  if (s == 1.2345) sink[0] = s;
  if (threadIdx.x == 0) { out[blockIdx.x * 4 + 2] = (int)((wall_clock64() - t0) / 100); }
}
template <int SCR, int NA = 24>
void run(hipStream_t st, int grid, size_t lds, const char* label) {
  int* d; double* sink;
  hipMalloc(&d, grid * 16); hipMalloc(&sink, 64); hipMemset(d, 0xff, grid * 16);
  hipFuncSetAttribute((const void*)k<SCR>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  int per = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)k<SCR>, 512, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  hipLaunchKernelGGL(k<SCR>, dim3(grid), dim3(512), lds, st, d, 300, sink);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  std::vector<int> h(grid * 4); hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
  std::map<int, int> cus; int perx[8] = {0};
  for (int bI = 0; bI < grid; ++bI) { const int hw = h[bI * 4 + 1]; const int cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (h[bI * 4] << 8); cus[cu]++; perx[h[bI * 4] & 7]++; }
  int hist[8] = {0}; for (auto& kv : cus) hist[kv.second < 7 ? kv.second : 7]++;
  int tmin = 0x7fffffff; for (int bI = 0; bI < grid; ++bI) tmin = std::min(tmin, h[bI * 4 + 3]);
  int first = 0; std::map<int, int> cus1;
  for (int bI = 0; bI < grid; ++bI) if (h[bI * 4 + 3] - tmin < 100 * 100) { ++first; const int hw = h[bI * 4 + 1]; cus1[((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (h[bI * 4] << 8)]++; }
  int h1[4] = {0}; for (auto& kv : cus1) h1[kv.second < 3 ? kv.second : 3]++;
  printf("    started within 100 us: %d workgroups on %zu CUs (CUs with 1 / 2 / 3+ of them: %d / %d / %d)\n", first, cus1.size(), h1[1], h1[2], h1[3]);
  printf("%-28s grid %d lds %zu occ-api %d: %.0f us (300 us spin => rounds %.1f); CUs %zu; CUs with 1/2/3/4+ WGs over the launch: %d/%d/%d/%d; per XCD %d %d %d %d %d %d %d %d\n", label, grid, lds, per, ms * 1e3, ms * 1e3 / 300.0,
         cus.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7], perx[0], perx[1], perx[2], perx[3], perx[4], perx[5], perx[6], perx[7]);
  hipFree(d); hipFree(sink);
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
  for (int b = 8; b < ncu; ++b) mask[b >> 5] |= 1u << (b & 31);
  hipStream_t sm, sp; hipExtStreamCreateWithCUMask(&sm, mask.size(), mask.data()); hipStreamCreateWithFlags(&sp, hipStreamNonBlocking);
  run<0>(sp, 2 * ncu, 73728, "plain stream, no scratch");
  run<0>(sm, 2 * (ncu - 8), 73728, "masked stream, no scratch");
  run<24>(sp, 2 * ncu, 73728, "plain stream, scratch");
  run<24>(sm, 2 * (ncu - 8), 73728, "masked stream, scratch");
  run<0>(sm, 2 * (ncu - 8), 65536, "masked, no scratch, 64K");
  run<24>(sm, 2 * (ncu - 8), 65536, "masked, scratch, 64K");
  run<24>(sm, 2 * (ncu - 8), 32768, "masked, scratch, 32K");
  run<0>(sm, ncu - 8, 73728, "masked, no scratch, 1 per CU");
  std::vector<uint32_t> m2((ncu + 31) / 32, 0u);
  for (int b = 16; b < ncu; ++b) m2[b >> 5] |= 1u << (b & 31);
  hipStream_t s2; hipExtStreamCreateWithCUMask(&s2, m2.size(), m2.data());
  run<0>(s2, 2 * (ncu - 16), 73728, "masked 16, no scratch");
  std::vector<uint32_t> m3((ncu + 31) / 32, 0u);
  for (int b = 0; b < ncu - 8; ++b) m3[b >> 5] |= 1u << (b & 31);
  hipStream_t s3; hipExtStreamCreateWithCUMask(&s3, m3.size(), m3.data());
  run<0>(s3, 2 * (ncu - 8), 73728, "masked last 8, no scratch");
  std::vector<uint32_t> m4((ncu + 31) / 32, 0xffffffffu);
  hipStream_t s4; hipExtStreamCreateWithCUMask(&s4, m4.size(), m4.data());
  run<0>(s4, 2 * ncu, 73728, "full mask, no scratch");
  return 0;
}
