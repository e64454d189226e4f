// Diagnostic (GPU box): serialized cost of tiny dependent kernels on the null stream, a created stream and a graph replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void spin(float* p, long long cycles) { long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (cycles < 0) p[0] = 1.f; }
__global__ void tiny_lds(float* p, int n) { __shared__ float sh[4096]; int i = blockIdx.x * blockDim.x + threadIdx.x; sh[threadIdx.x] = p[i % n]; __syncthreads(); if (i < n) p[i] = sh[255 - threadIdx.x] + 1.f; }
struct Big { float* p; int n; int pad[200]; };
__global__ void tiny_big(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < b.n) b.p[i] += 1.f; }
static double run(hipStream_t s, float* d, int iters, int blocks, bool big) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  Big b; b.p = d; b.n = blocks * 256;
  for (int i = 0; i < 20; ++i) { if (big) tiny_big<<<blocks, 256, 0, s>>>(b); else tiny<<<blocks, 256, 0, s>>>(d, blocks * 256); }
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i) { if (big) tiny_big<<<blocks, 256, 0, s>>>(b); else tiny<<<blocks, 256, 0, s>>>(d, blocks * 256); }
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / iters;
}
int main() {
  float* d; hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  for (int blocks : {1, 48, 512}) {
    printf("blocks=%d null: %.2f us  created: %.2f us  nonblocking: %.2f us  | big-arg null %.2f nonblocking %.2f\n", blocks,
           run(0, d, 2000, blocks, false), run(s1, d, 2000, blocks, false), run(s2, d, 2000, blocks, false),
           run(0, d, 2000, blocks, true), run(s2, d, 2000, blocks, true));
  }
  // memset between kernels
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (hipStream_t s : {(hipStream_t)0, s2}) {
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      for (int i = 0; i < 1000; ++i) { hipMemsetAsync(d, 0, 4096, s); tiny<<<1, 256, 0, s>>>(d, 256); }
      hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("memset+kernel pair on %s: %.2f us per pair\n", s ? "nonblocking" : "null", ms * 1e3 / 1000);
    }
  }
  // graph replay of 50 dependent tiny kernels
  {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 50; ++i) tiny<<<1, 256, 0, s2>>>(d, 256);
    hipStreamEndCapture(s2, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s2);
    hipStreamSynchronize(s2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s2);
    for (int i = 0; i < 40; ++i) hipGraphLaunch(ge, s2);
    hipEventRecord(e1, s2); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("graph of 50 tiny kernels: %.2f us per kernel\n", ms * 1e3 / (40 * 50));
  }
  // GPU-side floor: queue filled behind a long kernel, so the host is not the limiter
  for (int big = 0; big < 2; ++big) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Big b; b.p = d; b.n = 256;
    spin<<<1, 64, 0, s2>>>(d, 3000000);
    hipEventRecord(e0, s2);
    for (int i = 0; i < 300; ++i) { if (big) tiny_big<<<1, 256, 0, s2>>>(b); else tiny<<<1, 256, 0, s2>>>(d, 256); }
    hipEventRecord(e1, s2); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("GPU-side floor behind a long kernel (%s args): %.2f us per kernel\n", big ? "816-B" : "small", ms * 1e3 / 300);
  }
  // GPU-side floor by kernel size / kind (queue pre-filled behind a long kernel)
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](const char* what, auto&& body, int n) {
      spin<<<1, 64, 0, s2>>>(d, 4000000);
      hipEventRecord(e0, s2);
      for (int i = 0; i < n; ++i) body(i);
      hipEventRecord(e1, s2); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("GPU-side %-44s %.2f us per op\n", what, ms * 1e3 / n);
    };
    timed("tiny, 1 block", [&](int) { tiny<<<1, 256, 0, s2>>>(d, 256); }, 300);
    timed("tiny, 48 blocks", [&](int) { tiny<<<48, 256, 0, s2>>>(d, 48 * 256); }, 300);
    timed("tiny, 512 blocks", [&](int) { tiny<<<512, 256, 0, s2>>>(d, 512 * 256); }, 300);
    timed("tiny, 4096 blocks (4 MB rw)", [&](int) { tiny<<<4096, 256, 0, s2>>>(d, 4096 * 256); }, 300);
    timed("alternating tiny / tiny_lds", [&](int i) { if (i & 1) tiny<<<48, 256, 0, s2>>>(d, 48 * 256); else tiny_lds<<<48, 256, 0, s2>>>(d, 48 * 256); }, 300);
    timed("memsetAsync 4 KB", [&](int) { hipMemsetAsync(d, 0, 4096, s2); }, 300);
    timed("memsetAsync 4 MB", [&](int) { hipMemsetAsync(d, 0, 4 << 20, s2); }, 300);
    timed("memset 4 KB + tiny alternating", [&](int i) { if (i & 1) hipMemsetAsync(d, 0, 4096, s2); else tiny<<<48, 256, 0, s2>>>(d, 48 * 256); }, 300);
    timed("tiny on NULL stream, 48 blocks", [&](int) { tiny<<<48, 256, 0, 0>>>(d, 48 * 256); }, 300);
  }
  // host launch cost alone
  {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) tiny<<<1, 256, 0, s2>>>(d, 256);
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s2);
    printf("host launch cost: %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
  }
  return 0;
}
