// How fast does one SIMD issue v_mfma_f64_4x4x4 in the shape of the update task's inner loop (per k-step: 12 doubles from LDS, 32 MFMAs in two groups
// of 16; 4 k-steps per chunk, one barrier per chunk), with 2 or 4 waves per SIMD, as two 512-thread workgroups or one 1024-thread workgroup,
// and with a hardware barrier, a per-half LDS-counter barrier or none?
// build: hipcc -O3 -w --offload-arch=gfx950 scripts/mfma_rate_probe.hip -o scripts/_bin/mfma_rate_probe ; run on the GPU box (TIMELINE=1 prints per-wave chunk starts)
// A half's time runs to its LAST wave's end: without barriers the older wave of a SIMD pair runs ahead (56 : 44) and its own time alone would
// claim 15.5 cycles per MFMA for a pair that really needs 19.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
constexpr int kLd = 132;   // doubles per LDS row (the kernel's padding)
template <int THREADS, int BAR, int PER_CU, int PIPE = 0, int STAG = 0, int PRIO = 0>
__global__ __launch_bounds__(THREADS, PER_CU) void k(double* out, long long* cyc, int chunks, int dma, const double* src, int every) {
  extern __shared__ double sm_all[];
  __shared__ unsigned hb[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 7, half = tid >> 9;
  constexpr int NST = BAR == 3 ? 4 : 2;
  double* sm = sm_all + half * (2 * 16 * 2 * kLd);   // ring of 2 stages x 16 columns x (A row + B row)
  for (int i = tid & 511; i < NST * 16 * 2 * kLd; i += 512) sm[i] = 1e-3 * (i % 97);
  if (tid < 8) hb[tid] = 0;
  __syncthreads();
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0;
  const int ar = (wave >> 1) * 32 + (lane & 3), br = (wave & 1) * 64 + (lane & 3);
  const int kl = lane >> 4;  // the k index inside a 4-step this lane feeds
  if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(1);
  const long long t0 = clock64(), w0 = wall_clock64();
  unsigned want = 0;
  long long twait = 0;
  for (int c = 0; c < chunks; ++c) {
    if (blockIdx.x == 0 && lane == 0 && c >= 1024 && c < 1088) cyc[4096 + (c - 1024) * 8 + wave] = clock64();
    long long tb0 = 0;
    if (BAR == 1 && (c % every) == 0) { tb0 = clock64(); __syncthreads(); twait += clock64() - tb0; if (STAG > 0 && wave >= 4) __builtin_amdgcn_s_sleep(STAG); }
    if (BAR == 2) {
      want += 8;
      if (lane == 0) __hip_atomic_fetch_add(&hb[half], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(&hb[half], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
    }
    const double* st = sm + (c % NST) * (16 * 2 * kLd);
    auto dma_chunk = [&](int cc) {
      double* dstg = sm + (cc % NST) * (16 * 2 * kLd);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = (r * 512 + (tid & 511)) * 2;
        const int row = e >> 7, col = e & 127;
        __builtin_amdgcn_global_load_lds(src + ((size_t)blockIdx.x * 64 + (cc & 31)) * 4096 + e, (__attribute__((address_space(3))) void*)(dstg + row * kLd + col - (lane * 2)), 16, 0, 0);
      }
    };
    if (BAR == 3) {
      // barrier-free ring of four slots: per-slot counters say "all eight waves' rows of the chunk have landed" and "all eight waves have read it"
      if (c == 0) { dma_chunk(0); dma_chunk(1); dma_chunk(2); }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // my rows of chunk c + 1 (and everything older) have landed
      if (lane == 0) {
        if (c == 0) __hip_atomic_fetch_add(&hb[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&hb[(c + 1) & 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (c > 0) { const unsigned w = 8u * ((c - 1) / 4 + 1); while (__hip_atomic_load(&hb[4 + ((c - 1) & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < w) __builtin_amdgcn_s_sleep(1); }
      dma_chunk(c + 3);
      { const unsigned w = 8u * (c / 4 + 1); while (__hip_atomic_load(&hb[c & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < w) __builtin_amdgcn_s_sleep(1); }
    } else if (dma) {   // the next chunk's operands: 4 x 16-byte loads per thread straight into the other stage (same count as the kernel's)
      double* dstg = sm + ((c + 1) & 1) * (16 * 2 * kLd);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = (r * 512 + (tid & 511)) * 2;
        if (e < 16 * 2 * 128) {
          const int row = e >> 7, col = e & 127;
          __builtin_amdgcn_global_load_lds(src + ((size_t)blockIdx.x * 64 + (c & 31)) * 4096 + e, (__attribute__((address_space(3))) void*)(dstg + row * kLd + col - (lane * 2)), 16, 0, 0);
        }
      }
    }
    if (PIPE == 0) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double* pa = st + (ks * 4 + kl) * 2 * kLd + ar;
      const double* pb = pa + kLd - ar + br;
      double b[4], a[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = pb[j * 4 + ((lane >> 2) & 3) * 16];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = pa[i * 4];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    } else {
      // register pipeline: the operands of group g + 1 (16 MFMAs) are requested before group g's MFMAs are issued
      double b[2][4], a[2][4];
      auto rd = [&](int g, int buf) {
        const int ks = g >> 1, hf = g & 1;
        const double* pa = st + (ks * 4 + kl) * 2 * kLd + ar;
        const double* pb = pa + kLd - ar + br;
        if (hf == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[ks & 1][j] = pb[j * 4 + ((lane >> 2) & 3) * 16];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a[buf][i] = pa[(hf * 4 + i) * 4];
      };
      rd(0, 0);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) rd(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[(g & 1) * 16 + i * 4 + j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[g & 1][i], b[(g >> 1) & 1][j], acc[(g & 1) * 16 + i * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (BAR == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) __hip_atomic_fetch_add(&hb[4 + (c & 3)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    else if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[(size_t)blockIdx.x * THREADS + tid] = s;
  // the half's time: from its first wave's start to its LAST wave's end (without barriers the older wave of a SIMD pair runs ahead and
  // finishes early; its own time says nothing about the pair's throughput)
  __shared__ long long s_t[2][4];
  if ((tid & 511) == 0) { s_t[half][0] = t0; s_t[half][1] = w0; s_t[half][2] = 0; s_t[half][3] = 0; }
  __syncthreads();
  if (lane == 0) { atomicMax((unsigned long long*)&s_t[half][2], (unsigned long long)t1); atomicMax((unsigned long long*)&s_t[half][3], (unsigned long long)w1); }
  __syncthreads();
  if ((tid & 511) == 0) { cyc[(blockIdx.x * 2 + half) * 2] = s_t[half][2] - s_t[half][0]; cyc[(blockIdx.x * 2 + half) * 2 + 1] = s_t[half][3] - s_t[half][1]; }
  if (tid == 64 * 5 && THREADS <= 512) cyc[(blockIdx.x * 2 + 1) * 2] = twait;
}
template <int THREADS, int BAR, int PER_CU, int PIPE = 0, int STAG = 0, int PRIO = 0>
void run(const char* label, int ncu, int chunks, int dma, size_t lds, int every = 1) {
  const int grid = ncu * PER_CU;
  double* out; long long* cyc; double* src;
  hipMalloc(&out, (size_t)grid * THREADS * 8); hipMalloc(&cyc, (grid * 4 + 8192) * 8); hipMemset(cyc, 0, (grid * 4 + 8192) * 8);
  hipMalloc(&src, (size_t)grid * 64 * 4096 * 8); hipMemset(src, 0, (size_t)grid * 64 * 4096 * 8);
  hipFuncSetAttribute((const void*)k<THREADS, BAR, PER_CU, PIPE, STAG, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<THREADS, BAR, PER_CU, PIPE, STAG, PRIO>), dim3(grid), dim3(THREADS), lds, 0, out, cyc, chunks, dma, src, every);
  hipDeviceSynchronize();
  std::vector<long long> h(grid * 4); hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
  std::vector<double> per, wall;
  for (int b = 0; b < grid; ++b) for (int hf = 0; hf < (THREADS > 512 ? 2 : 1); ++hf) { per.push_back((double)h[(b * 2 + hf) * 2]); wall.push_back((double)h[(b * 2 + hf) * 2 + 1]); }
  std::sort(per.begin(), per.end()); std::sort(wall.begin(), wall.end());
  const double med = per[per.size() / 2], wmed = wall[wall.size() / 2];
  const int waves_per_simd = THREADS >= 256 ? THREADS / 256 * PER_CU : 1;
  // every half (8 waves, 2 per SIMD) issues chunks*128 MFMAs per wave; a SIMD therefore issues waves_per_simd*chunks*128 in `med` cycles
  if (THREADS <= 512 && BAR == 1) { std::vector<double> tw; for (int b = 0; b < grid; ++b) tw.push_back((double)h[(b * 2 + 1) * 2]); std::sort(tw.begin(), tw.end()); printf("    wave 5 waits %.0f cycles per barrier (median)\n", tw[tw.size() / 2] / (chunks / every)); }
  printf("%-44s waves/SIMD %d dma %d: %.0f cycles per chunk per half, %.1f cycles per MFMA per SIMD  (clock %.2f GHz, %s)\n", label, waves_per_simd, dma, med / chunks,
         med / ((double)waves_per_simd * chunks * 128), med / (wmed * 10.0), hipGetErrorString(hipGetLastError()));
  if (THREADS == 512 && PER_CU == 1 && getenv("TIMELINE")) {
    std::vector<long long> tl(512); hipMemcpy(tl.data(), cyc + 4096, 512 * 8, hipMemcpyDeviceToHost);
    for (int c = 0; c < 40; ++c) { printf("      chunk %2d: start of waves 0..7 relative to wave 0 of chunk 0:", c); for (int w = 0; w < 8; ++w) printf(" %6lld", tl[c * 8 + w] - tl[0]); printf("\n"); }
  }
  hipFree(out); hipFree(cyc); hipFree(src);
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount, chunks = 2000;
  run<512, 1, 1>("512, s_barrier every 16 columns", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1>("512, s_barrier every 32 columns", ncu, chunks, 0, 140 * 1024, 2);
  run<512, 1, 1>("512, s_barrier every 64 columns", ncu, chunks, 0, 140 * 1024, 4);
  run<512, 1, 1>("512, s_barrier every 256 columns", ncu, chunks, 0, 140 * 1024, 16);
  run<512, 1, 1, 0, 1>("512, s_barrier, waves 4-7 sleep 1 after it", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1, 0, 2>("512, s_barrier, waves 4-7 sleep 2 after it", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1, 0, 4>("512, s_barrier, waves 4-7 sleep 4 after it", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1, 0, 8>("512, s_barrier, waves 4-7 sleep 8 after it", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1, 0, 4>("512, s_barrier / 32 columns, waves 4-7 sleep 4", ncu, chunks, 0, 140 * 1024, 2);
  run<512, 1, 1, 0, 8>("512, s_barrier / 32 columns, waves 4-7 sleep 8", ncu, chunks, 0, 140 * 1024, 2);
  run<512, 1, 1, 1, 4>("512, s_barrier, sleep 4, register pipeline", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1, 0, 0, 1>("512, s_barrier, waves 4-7 at priority 1", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1, 0, 0, 2>("512, s_barrier, waves 0-3 at priority 1", ncu, chunks, 0, 140 * 1024, 1);
  run<512, 1, 1, 1, 0, 1>("512, s_barrier, waves 4-7 at priority 1, register pipeline", ncu, chunks, 0, 140 * 1024, 1);
  run<256, 1, 1>("256 (one wave per SIMD), s_barrier every 16", ncu, chunks, 0, 140 * 1024, 1);
  run<256, 0, 1>("256 (one wave per SIMD), no barrier", ncu, chunks, 0, 140 * 1024, 1);
  run<256, 0, 1, 1>("256 (one wave per SIMD), no barrier, register pipeline", ncu, chunks, 0, 140 * 1024, 1);
  for (int dma = 0; dma < 2; ++dma) {
    run<512, 1, 1>("one 512-thread WG per CU, s_barrier", ncu, chunks, dma, 140 * 1024);
    run<512, 0, 1>("one 512-thread WG per CU, no barrier", ncu, chunks, dma, 140 * 1024);
    run<512, 1, 2>("two 512-thread WGs per CU, s_barrier", ncu, chunks, dma, 70 * 1024);
    run<512, 0, 2>("two 512-thread WGs per CU, no barrier", ncu, chunks, dma, 70 * 1024);
    run<1024, 1, 1>("one 1024-thread WG per CU, s_barrier", ncu, chunks, dma, 140 * 1024);
    run<1024, 2, 1>("one 1024-thread WG per CU, half barriers", ncu, chunks, dma, 140 * 1024);
    run<1024, 0, 1>("one 1024-thread WG per CU, no barrier", ncu, chunks, dma, 140 * 1024);
    if (dma) run<512, 3, 1, 0>("one 512 WG, four-slot ring without barriers", ncu, chunks, dma, 140 * 1024);
    if (dma) run<512, 3, 1, 1>("one 512 WG, four-slot ring, register pipeline", ncu, chunks, dma, 140 * 1024);
    run<512, 1, 1, 1>("one 512 WG, s_barrier, register pipeline", ncu, chunks, dma, 140 * 1024);
    run<512, 0, 1, 1>("one 512 WG, no barrier, register pipeline", ncu, chunks, dma, 140 * 1024);
    run<1024, 2, 1, 1>("one 1024 WG, half barriers, register pipeline", ncu, chunks, dma, 140 * 1024);
  }
  return 0;
}
