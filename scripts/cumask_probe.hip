// Probe: (1) which CUs a CU-masked stream uses; (2) whether a 159-KB-LDS single-workgroup kernel on a second
// stream can start while a long 2-WG/CU kernel saturates the masked stream.
// build: hipcc --offload-arch=gfx950 -O2 scripts/cumask_probe.hip -o gpurun_out/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned hwid() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); }
__device__ inline unsigned xccid() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); }

__device__ long long g_first = 0x7fffffffffffffffLL, g_last = 0;
__global__ void k_where(unsigned* hist, long long spin) {
  extern __shared__ double lds[];
  if (spin > 5000) spin = spin / 2 + (long long)((blockIdx.x * 2654435761u) >> 16) % spin;   // desynchronised durations
  if (threadIdx.x == 0) {
    atomicMin((unsigned long long*)&g_first, (unsigned long long)wall_clock64());
    unsigned h = hwid(), x = xccid() & 15;
    unsigned cu = (h >> 8) & 15, sh = (h >> 12) & 1, se = (h >> 13) & 7;
    unsigned id = ((x * 8 + se) * 2 + sh) * 16 + cu;
    atomicAdd(&hist[id], 1u);
    lds[0] = (double)h;
  }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0) atomicMax((unsigned long long*)&g_last, (unsigned long long)wall_clock64());
}

__global__ void k_big_lds(long long* out, long long spin) {
  extern __shared__ double lds[];
  long long t0 = wall_clock64();
  if (threadIdx.x == 0) { lds[0] = 1.0; out[0] = t0; out[2] = ((long long)(xccid() & 15) << 32) | hwid(); }
  while (wall_clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0) out[1] = wall_clock64();
}

int main(int argc, char** argv) {
  const int nres = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("CUs %d  wallclock kHz %d\n", pr.multiProcessorCount, pr.clockRate);
  int wcr = 0; CK(hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0)); printf("wall clock rate kHz %d\n", wcr);
  const int NID = 16 * 8 * 2 * 16;
  unsigned* hist; CK(hipMalloc(&hist, NID * 4));
  std::vector<unsigned> h(NID);
  auto show = [&](const char* tag) {
    CK(hipMemcpy(h.data(), hist, NID * 4, hipMemcpyDeviceToHost));
    int used = 0; for (int i = 0; i < NID; ++i) used += h[i] != 0;
    printf("%s: %d distinct (xcc,se,sh,cu) used\n", tag, used);
    for (int x = 0; x < 8; ++x) { printf("  xcc%d:", x); for (int i = 0; i < 256; ++i) { int id = x * 256 + i; if (h[id]) printf(" %d.%d.%d", (i >> 5) & 7, (i >> 4) & 1, i & 15); } printf("\n"); }
  };
  CK(hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_big_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t full; CK(hipStreamCreate(&full));
  CK(hipMemset(hist, 0, NID * 4));
  hipLaunchKernelGGL(k_where, dim3(4096), dim3(256), 74 * 1024, full, hist, 2000LL);
  CK(hipStreamSynchronize(full)); show("full stream");

  for (int variant = 0; variant < 3; ++variant) {
    std::vector<uint32_t> mask(8, 0xffffffffu);
    if (variant == 0) mask[0] &= ~0x1u;          // clear bit 0
    if (variant == 1) mask[0] &= ~((1u << nres) - 1u);   // clear bits 0..nres-1
    if (variant == 2) mask[0] &= ~0x101u;        // clear bits 0 and 8
    hipStream_t ms; hipError_t e = hipExtStreamCreateWithCUMask(&ms, 8, mask.data());
    printf("variant %d: hipExtStreamCreateWithCUMask -> %s\n", variant, hipGetErrorString(e));
    if (e != hipSuccess) continue;
    CK(hipMemset(hist, 0, NID * 4));
    hipLaunchKernelGGL(k_where, dim3(8192), dim3(256), 74 * 1024, ms, hist, 2000LL);
    CK(hipStreamSynchronize(ms)); show("masked");
    if (variant == 1) {
      // concurrency: saturate ms for ~20 ms (8192 WGs x 100us / 510 slots ~ 1.6 ms per wave ... use longer spin)
      long long* out; CK(hipMalloc(&out, 128)); CK(hipMemset(out, 0, 128));
      int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      hipStream_t pb; CK(hipStreamCreateWithPriority(&pb, hipStreamNonBlocking, hi));
      for (int which = 0; which < 2; ++which) {
        hipStream_t sa = which == 0 ? ms : full;
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        CK(hipDeviceSynchronize());
        { long long big = 0x7fffffffffffffffLL, z = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_first), &big, 8)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_last), &z, 8)); }
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(k_where, dim3(16384), dim3(256), 74 * 1024, sa, hist, 10000LL);   // 100 us per WG at 100 MHz
        CK(hipEventRecord(e1, sa));
        auto t0 = std::chrono::steady_clock::now();
        // let it start
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.0005) {}
        for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k_big_lds, dim3(1), dim3(256), 159 * 1024, pb, out + rep * 3, 5000LL);
        CK(hipEventRecord(e2, pb));
        CK(hipEventSynchronize(e2));
        double tb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CK(hipEventSynchronize(e1));
        double ta = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        float msA; CK(hipEventElapsedTime(&msA, e0, e1));
        long long ho[12]; CK(hipMemcpy(ho, out, 96, hipMemcpyDeviceToHost));
        printf("saturating stream %s: big-LDS chain of 4 done after %.3f ms (host), saturating kernel done after %.3f ms (event %.3f ms)\n",
               which == 0 ? "MASKED" : "FULL", tb * 1e3, ta * 1e3, msA);
        long long gf, gl; CK(hipMemcpyFromSymbol(&gf, HIP_SYMBOL(g_first), 8)); CK(hipMemcpyFromSymbol(&gl, HIP_SYMBOL(g_last), 8));
        printf("   saturating kernel: first WG at %lld, last WG end at %lld (ticks rel. to first big-LDS start)\n", gf - ho[0], gl - ho[0]);
        for (int rep = 0; rep < 4; ++rep) printf("   rep %d: start %lld end %lld (ticks)  xcc %lld hwid cu %lld se %lld\n", rep, ho[rep * 3] - ho[0], ho[rep * 3 + 1] - ho[0], ho[rep * 3 + 2] >> 32, (ho[rep * 3 + 2] >> 8) & 15, (ho[rep * 3 + 2] >> 13) & 7);
      }
    }
    CK(hipStreamDestroy(ms));
  }
  return 0;
}
