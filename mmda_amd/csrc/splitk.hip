// The reduce launch of the deterministic split-K (splitk.h): one thread per output element (n fastest: the slab reads and the C
// accesses of a wave are contiguous), slabs summed in slice order 0, 1, ..., sk - 1.
#include "splitk.h"

namespace {

struct SplitKLaunch {
  SplitKJob j[SPLITK_JOBS_MAX];
  int start[SPLITK_JOBS_MAX + 1];      // first block of each job
  int n;
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitKLaunch L) {
  int ji = 0;
#pragma unroll
  for (int k = 1; k < SPLITK_JOBS_MAX; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) ji = k;
  const SplitKJob& J = L.j[ji];
  const int Ne = J.N + (J.bias_grad ? 1 : 0);
  const int64_t per = (int64_t)J.M * Ne;
  const int64_t e = (int64_t)(blockIdx.x - L.start[ji]) * 256 + threadIdx.x;
  if (e >= per * J.batch) return;
  const int bz = (int)(e / per);
  const int64_t r = e - (int64_t)bz * per;
  const int m = (int)(r / Ne), n = (int)(r - (int64_t)m * Ne);
  const int64_t slice = (int64_t)J.M * J.ldn;
  const float* p = J.slab + (int64_t)bz * J.sk * slice + (int64_t)m * J.ldn + n;
  float v = p[0];
  for (int s = 1; s < J.sk; ++s) v += p[(int64_t)s * slice];
  const int mo = J.perm_m_H > 0 ? splitk_orig(m, J.perm_m_H) : m;
  if (n == J.N) {                                            // the virtual ones-column: bias gradient(s), single writer per entry
    J.bias_grad[bz * J.strideBias + mo] += v;
    if (J.bias_grad2) J.bias_grad2[bz * J.strideBias + mo] += v;
    return;
  }
  const float alpha = J.alpha == 0.f ? 1.f : J.alpha;
  float out = alpha * v;
  const int nb = J.perm_n_H > 0 ? splitk_orig(n, J.perm_n_H) : n;
  if (J.bias) out += J.bias[bz * J.strideBias + nb];
  if (J.bias2) out += J.bias2[bz * J.strideBias + nb];
  float* dst = J.C + bz * J.strideC + (int64_t)mo * J.ldc + n;
  if (J.accumulate) out += *dst;
  *dst = out;
}

}  // namespace

int mmda_splitk_reduce(const SplitKJob* jobs, int n, hipStream_t s) {
  for (int base = 0; base < n; base += SPLITK_JOBS_MAX) {
    const int cnt = (n - base) < SPLITK_JOBS_MAX ? (n - base) : SPLITK_JOBS_MAX;
    SplitKLaunch L;
    int blocks = 0;
    L.n = cnt;
    for (int i = 0; i < cnt; ++i) {
      const SplitKJob& J = jobs[base + i];
      L.j[i] = J;
      L.start[i] = blocks;
      const int64_t elems = (int64_t)J.M * (J.N + (J.bias_grad ? 1 : 0)) * J.batch;
      blocks += (int)((elems + 255) / 256);
    }
    for (int i = cnt; i <= SPLITK_JOBS_MAX; ++i) L.start[i] = blocks;
    for (int i = cnt; i < SPLITK_JOBS_MAX; ++i) L.j[i] = L.j[0];
    if (blocks == 0) continue;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, L);
    MMDA_CHECK_LAUNCH("mmda_splitk_reduce");
  }
  return MMDA_OK;
}
