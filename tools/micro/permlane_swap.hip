#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  unsigned x = threadIdx.x * 10;
  u2 a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  u2 b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[threadIdx.x * 4 + 0] = a[0]; out[threadIdx.x * 4 + 1] = a[1]; out[threadIdx.x * 4 + 2] = b[0]; out[threadIdx.x * 4 + 3] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 16); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int l : {0, 5, 16, 21, 32, 37, 48, 53}) printf("lane %2d: p16 (%u, %u)  p32 (%u, %u)\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
