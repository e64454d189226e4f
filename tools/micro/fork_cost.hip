// What does a stream fork / join cost the main stream?  main: K1 -> [fork] -> K2 -> [join] -> (next iteration); side: K3 between fork and join.
// Variants of the fork: hipEventRecord on the main stream (a marker packet behind K1) vs the event bound to K1's own dispatch
// (hipExtLaunchKernelGGL stopEvent); same for the join on the side stream.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/fork_cost tools/micro/fork_cost.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 12345) *sink = 1;
}

int main() {
  hipStream_t main_s, side_s;
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side_s, hipStreamNonBlocking));
  hipEvent_t ef, ej;
  for (int flavour = 0; flavour < 3; ++flavour) {
  const unsigned fl = hipEventDisableTiming | (flavour == 1 ? hipEventDisableSystemFence : flavour == 2 ? hipEventReleaseToDevice : 0u);
  printf("---- events created with hipEventDisableTiming%s\n", flavour == 1 ? " | hipEventDisableSystemFence" : flavour == 2 ? " | hipEventReleaseToDevice" : "");
  CK(hipEventCreateWithFlags(&ef, fl));
  CK(hipEventCreateWithFlags(&ej, fl));
  const long long T10 = 1000;     // wall_clock64 runs at 100 MHz: 10 us
  const int N = 300;
  int* sink = nullptr;
  for (int variant = 0; variant < 6; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        switch (variant) {
          case 0:   // no fork: K1, K2 back to back
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            break;
          case 1:   // event fork, no join inside the loop (the side stream simply runs behind)
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            CK(hipEventRecord(ef, main_s)); CK(hipStreamWaitEvent(side_s, ef, 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, side_s, T10 / 2, sink);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            break;
          case 2:   // fork bound to K1's dispatch
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, nullptr, ef, 0, T10, sink);
            CK(hipStreamWaitEvent(side_s, ef, 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, side_s, T10 / 2, sink);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            break;
          case 3:   // event fork + event join (side kernel shorter than K2: the join never really waits)
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            CK(hipEventRecord(ef, main_s)); CK(hipStreamWaitEvent(side_s, ef, 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, side_s, T10 / 2, sink);
            CK(hipEventRecord(ej, side_s));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            CK(hipStreamWaitEvent(main_s, ej, 0));
            break;
          case 4:   // both bound to dispatches
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, nullptr, ef, 0, T10, sink);
            CK(hipStreamWaitEvent(side_s, ef, 0));
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, side_s, nullptr, ej, 0, T10 / 2, sink);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            CK(hipStreamWaitEvent(main_s, ej, 0));
            break;
          case 5:   // event fork, join bound to the side kernel's dispatch
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            CK(hipEventRecord(ef, main_s)); CK(hipStreamWaitEvent(side_s, ef, 0));
            hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, side_s, nullptr, ej, 0, T10 / 2, sink);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, main_s, T10, sink);
            CK(hipStreamWaitEvent(main_s, ej, 0));
            break;
        }
      }
      CK(hipStreamSynchronize(main_s));
      CK(hipStreamSynchronize(side_s));
      auto t1 = std::chrono::steady_clock::now();
      const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
      const char* names[6] = {"no fork (K1 K2)", "event fork", "ext-launch fork", "event fork + event join", "ext fork + ext join", "event fork + ext join"};
      if (rep == 1) printf("%-28s %7.2f us per iteration (2 x 10 us of kernels on the main stream)\n", names[variant], us);
    }
  }
  CK(hipEventDestroy(ef)); CK(hipEventDestroy(ej));
  }
  return 0;
}
