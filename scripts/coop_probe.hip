// Decides the design of the persistent solve kernels:
//  (1) does hipLaunchCooperativeKernel work here, and how many 256-thread workgroups are co-resident?
//  (2) what does a grid-wide barrier cost (cooperative_groups grid.sync() vs a hand-written atomic barrier)?
//  (3) which HBM rate does a column-major panel GEMV (w[r] -= sum_c L[r, c] y[c]) reach from inside such a kernel,
//      with the rows x columns panel cut 2-D over all workgroups (partial sums in slots)?
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned nwg, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++epoch;
    __threadfence();
    const unsigned target = epoch * nwg;
    atomicAdd(bar, 1u);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_bar_cg(int iters, double* out) {
  cg::grid_group g = cg::this_grid();
  double v = 0;
  for (int i = 0; i < iters; ++i) { v += 1.0; g.sync(); }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v;
}
__global__ __launch_bounds__(256) void k_bar_own(int iters, unsigned* bar, double* out) {
  unsigned epoch = 0;
  double v = 0;
  for (int i = 0; i < iters; ++i) { v += 1.0; grid_barrier(bar, gridDim.x, epoch); }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v;
}

// panel GEMV inside a persistent kernel: `steps` dependent panels, each rows x cols (column-major, ld = ldl), barrier between
// them.  Work item = (64-row chunk, column split): wave wv of the workgroup takes a quarter of the split's columns.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_gemv(const double* __restrict__ L, int ldl, int rows, int cols, int csplit, const double* __restrict__ y,
                                              double* __restrict__ part, int steps, unsigned* bar) {
  __shared__ double ys[1024];
  __shared__ double red[4][64];
  unsigned epoch = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nchunk = (rows + 63) / 64;
  const int nitem = nchunk * csplit;
  const int cw = cols / csplit;           // columns per split (multiple of 4 * UNROLL)
  for (int s = 0; s < steps; ++s) {
    for (int it = blockIdx.x; it < nitem; it += gridDim.x) {
      const int ch = it % nchunk, sp = it / nchunk;
      const int r = min(ch * 64 + lane, rows - 1);
      const int c0 = sp * cw + wv * (cw / 4);
      for (int c = threadIdx.x; c < cw; c += 256) ys[c] = y[sp * cw + c];
      __syncthreads();
      const double* Lp = L + (size_t)c0 * ldl + r;
      double a = 0;
      for (int c = 0; c < cw / 4; c += UNROLL) {
        double v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = Lp[(size_t)(c + u) * ldl];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) a += v[u] * ys[wv * (cw / 4) + c + u];
      }
      red[wv][lane] = a;
      __syncthreads();
      if (wv == 0) part[(size_t)sp * rows + r] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
      __syncthreads();
    }
    if (steps > 1) grid_barrier(bar, gridDim.x, epoch);
  }
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("device %s  CUs %d  cooperativeLaunch %d\n", prop.name, prop.multiProcessorCount, prop.cooperativeLaunch);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bar_own, 256, 0);
  printf("occupancy k_bar_own: %d workgroups / CU\n", occ);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gemv<32>, 256, 0);
  printf("occupancy k_gemv<32>: %d workgroups / CU\n", occ);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  double* d_out; hipMalloc(&d_out, 64);
  unsigned* bar; hipMalloc(&bar, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nwg : {256, 512, 1024}) {
    int iters = 200;
    void* a1[] = {&iters, &d_out};
    hipError_t er = hipLaunchCooperativeKernel((void*)k_bar_cg, dim3(nwg), dim3(256), a1, 0, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    er = hipLaunchCooperativeKernel((void*)k_bar_cg, dim3(nwg), dim3(256), a1, 0, st);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("grid.sync()  %4d workgroups: %s  %.2f us per barrier\n", nwg, hipGetErrorString(er), ms * 1e3 / iters);
    hipMemsetAsync(bar, 0, 64, st);
    void* a2[] = {&iters, &bar, &d_out};
    er = hipLaunchCooperativeKernel((void*)k_bar_own, dim3(nwg), dim3(256), a2, 0, st);
    hipStreamSynchronize(st);
    hipMemsetAsync(bar, 0, 64, st);
    hipEventRecord(e0, st);
    er = hipLaunchCooperativeKernel((void*)k_bar_own, dim3(nwg), dim3(256), a2, 0, st);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    hipEventElapsedTime(&ms, e0, e1);
    printf("own barrier  %4d workgroups: %s  %.2f us per barrier\n", nwg, hipGetErrorString(er), ms * 1e3 / iters);
    // plain (non-cooperative) launch of the own barrier with a grid that fits: does it also complete?
    if (nwg <= 512) {
      hipMemsetAsync(bar, 0, 64, st);
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(k_bar_own, dim3(nwg), dim3(256), 0, st, iters, bar, d_out);
      hipEventRecord(e1, st); hipStreamSynchronize(st);
      hipEventElapsedTime(&ms, e0, e1);
      printf("own barrier, plain launch %4d workgroups: %.2f us per barrier\n", nwg, ms * 1e3 / iters);
    }
  }
  // GEMV rates
  const int ldl = 16640;
  double* L; hipMalloc(&L, (size_t)ldl * 1024 * 8 * 2);
  hipMemset(L, 0, (size_t)ldl * 1024 * 8 * 2);
  double *y, *part; hipMalloc(&y, 1024 * 8); hipMalloc(&part, (size_t)16 * ldl * 8);
  hipMemset(y, 0, 1024 * 8);
  for (int rows : {16000, 8000, 4000, 2000, 1000}) {
    for (int csplit : {1, 2, 4, 8}) {
      for (int nwg : {256, 512, 1024}) {
        int cols = 1024, steps = 1;
        void* a3[] = {(void*)&L, (void*)&ldl, &rows, &cols, &csplit, &y, &part, &steps, &bar};
        hipMemsetAsync(bar, 0, 64, st);
        hipLaunchCooperativeKernel((void*)k_gemv<32>, dim3(nwg), dim3(256), a3, 0, st);
        hipStreamSynchronize(st);
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0, st);
          hipLaunchCooperativeKernel((void*)k_gemv<32>, dim3(nwg), dim3(256), a3, 0, st);
          hipEventRecord(e1, st); hipStreamSynchronize(st);
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        printf("gemv rows %5d x 1024, csplit %d, %4d wgs: %.1f us  %.2f TB/s\n", rows, csplit, nwg, best * 1e3, (double)rows * 1024 * 8 / (best * 1e-3) / 1e12);
      }
    }
  }
  // a sweep-like sequence: 16 dependent panels with barriers, rows shrinking is approximated by fixed 8000 rows
  for (int csplit : {2, 4}) {
    int rows = 8000, cols = 1024, steps = 16, nwg = 512;
    void* a3[] = {(void*)&L, (void*)&ldl, &rows, &cols, &csplit, &y, &part, &steps, &bar};
    hipMemsetAsync(bar, 0, 64, st);
    hipLaunchCooperativeKernel((void*)k_gemv<32>, dim3(nwg), dim3(256), a3, 0, st);
    hipStreamSynchronize(st);
    hipMemsetAsync(bar, 0, 64, st);
    hipEventRecord(e0, st);
    hipLaunchCooperativeKernel((void*)k_gemv<32>, dim3(nwg), dim3(256), a3, 0, st);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("16 dependent panels of 8000 x 1024 (csplit %d) in one kernel: %.1f us total, %.2f TB/s\n", csplit, ms * 1e3, 16.0 * rows * 1024 * 8 / (ms * 1e-3) / 1e12);
  }
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
