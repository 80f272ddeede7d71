// Is a captured hipGraph of dependent small kernels faster than the same kernels launched on a stream?
// (decides whether the launch-bound part of small factorisations is worth capturing)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(double* x, int iters) {
  double v = x[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
  x[threadIdx.x] = v;
}
int main() {
  double* d; hipMalloc(&d, 1024 * 8); hipMemset(d, 0, 1024 * 8);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int iters : {200, 4000, 20000}) {      // ~1 us, ~10 us, ~45 us kernels
    const int N = 600;
    auto run_stream = [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_spin, dim3(1 + (i % 3) * 40), dim3(256), 0, st, d, iters); };
    run_stream(); hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) run_stream();
    hipStreamSynchronize(st);
    double ts = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 5;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    run_stream();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double tg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 5;
    printf("iters %6d: %d dependent kernels  stream %.1f us (%.2f us each)   graph %.1f us (%.2f us each)\n", iters, N, ts * 1e6, ts * 1e6 / N, tg * 1e6, tg * 1e6 / N);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
