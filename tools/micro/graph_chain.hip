// Micro-benchmark: does a hipGraph shorten the GPU-side gap between dependent small kernels?
// 40 dependent kernels (each ~2 us of work on one workgroup), launched (a) on a stream, (b) as a captured graph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
int main() {
  float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 40, REP = 200;
  for (int iters : {10, 400}) {
    for (int w = 0; w < 3; ++w) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, iters); }
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < REP; ++r) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, iters);
    hipStreamSynchronize(s);
    double us_stream = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (REP * N);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, iters);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (REP * N);
    printf("iters=%d: per kernel  stream %.2f us   graph %.2f us\n", iters, us_stream, us_graph);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
