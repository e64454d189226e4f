// Which XCD does block b run on?  (HW_REG_XCC_ID, the register the recurrent kernels' placement check reads.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(6164) & 15u;
}
int main() {
  unsigned* d; unsigned h[64];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int i = 0; i < 64; ++i) printf("%u%s", h[i], (i % 8 == 7) ? "\n" : " ");
  return 0;
}
