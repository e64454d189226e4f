// Diagnostic (GPU box): does cycling through DIFFERENT kernels (cold instruction fetch) raise the per-kernel floor?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(float* p, long long cycles) { long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (cycles < 0) p[0] = 1.f; }
// straight-line body of ~N*8 VALU instructions that the compiler cannot fold (depends on loaded data and ID)
template <int ID, int N>
__global__ void body(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float x = p[i % n], y = (float)ID;
#pragma unroll
  for (int k = 0; k < N; ++k) { x = x * 1.0001f + y; y = y * 0.9999f + x * (float)(k + ID); x = __builtin_fmaf(x, y, (float)k); y = y - x * 0.5f; }
  if (i < n) p[i] = x + y;
}
template <int N, int... IDs>
static void launch_cycle(int which, float* d, int blocks, hipStream_t s) {
  int k = 0;
  ((which == k++ ? (void)(body<IDs, N><<<blocks, 256, 0, s>>>(d, blocks * 256)) : (void)0), ...);
}
template <int N>
static void test(float* d, hipStream_t s, int blocks, int distinct) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 32; ++i) launch_cycle<N, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15>(i % 16, d, blocks, s);
  (void)hipStreamSynchronize(s);
  spin<<<1, 64, 0, s>>>(d, 5000000);
  (void)hipEventRecord(e0, s);
  const int n = 320;
  for (int i = 0; i < n; ++i) launch_cycle<N, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15>(i % distinct, d, blocks, s);
  (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("body of ~%5d instr, %4d blocks, cycling %2d distinct kernels: %.2f us per kernel\n", N * 8, blocks, distinct, ms * 1e3 / n);
}
int main() {
  float* d; (void)hipMalloc(&d, 1 << 24); (void)hipMemset(d, 0, 1 << 24);
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int blocks : {2, 48, 512}) {
    for (int distinct : {1, 2, 16}) test<16>(d, s, blocks, distinct);
    for (int distinct : {1, 2, 16}) test<256>(d, s, blocks, distinct);
    for (int distinct : {1, 2, 16}) test<1024>(d, s, blocks, distinct);
  }
  return 0;
}
