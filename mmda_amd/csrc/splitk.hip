// The reduce launch of the deterministic split-K (splitk.h): one thread per output element (n fastest: the slab reads and the C
// accesses of a wave are contiguous), slabs summed in slice order 0, 1, ..., sk - 1.
#include "splitk.h"

namespace {

struct SplitKLaunch {
  SplitKJob j[SPLITK_JOBS_MAX];
  int start[SPLITK_JOBS_MAX + 1];      // first block of each job
  int n;
};

// 16-byte form: ldn, N and ldc multiples of four, C 16-byte aligned (the LSTM-sized problems): a thread sums four neighbouring columns.
// The slab rows are ldn wide; the group that starts at column N holds the bias-gradient column first and padding behind it.
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(SplitKLaunch L) {
  int ji = 0;
#pragma unroll
  for (int k = 1; k < SPLITK_JOBS_MAX; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) ji = k;
  const SplitKJob& J = L.j[ji];
  const int G = J.ldn >> 2;                                  // column groups per row
  const int64_t e = (int64_t)(blockIdx.x - L.start[ji]) * 256 + threadIdx.x;
  if (e >= (int64_t)J.M * G) return;
  const int m = (int)(e / G), n = (int)(e - (int64_t)m * G) * 4;
  const int64_t slice = (int64_t)J.M * J.ldn;
  const f32x4* p = reinterpret_cast<const f32x4*>(J.slab + (int64_t)m * J.ldn + n);
  f32x4 v = p[0];
  for (int s = 1; s < J.sk; ++s) v += p[(int64_t)s * (slice >> 2)];
  const int mo = J.perm_m_H > 0 ? splitk_orig(m, J.perm_m_H) : m;
  if (n >= J.N) {                                            // bias-gradient column (n == N) + padding
    if (J.bias_grad && n == J.N) {
      J.bias_grad[mo] += v[0];
      if (J.bias_grad2) J.bias_grad2[mo] += v[0];
    }
    return;
  }
  const float alpha = J.alpha == 0.f ? 1.f : J.alpha;
  f32x4 out = alpha * v;
  if (J.bias || J.bias2) {
    // n is a multiple of 4: with the gate interleave the four columns are the four gates of ONE unit, orig = nb0 + e * H
    int nb0 = n, nbs = 1;
    if (J.perm_n_H > 0) { const int Gt = 4 * J.perm_n_H, d = n / Gt; nb0 = d * Gt + ((n - d * Gt) >> 2); nbs = J.perm_n_H; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (J.bias) out[q] += J.bias[nb0 + q * nbs];
      if (J.bias2) out[q] += J.bias2[nb0 + q * nbs];
    }
  }
  f32x4* dst = reinterpret_cast<f32x4*>(J.C + (int64_t)mo * J.ldc + n);
  if (J.accumulate) out += *dst;
  *dst = out;
}

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
  // jobs that qualify for the 16-byte form go out in launches of their own
  auto is_vec = [](const SplitKJob& J) {
    return J.batch == 1 && !(J.ldn & 3) && !(J.N & 3) && !(J.ldc & 3) && !((uintptr_t)J.C & 15) && !((uintptr_t)J.slab & 15);
  };
  for (int form = 0; form < 2; ++form) {
    SplitKLaunch L;
    int blocks = 0;
    L.n = 0;
    auto flush = [&]() -> int {
      if (L.n == 0 || blocks == 0) { L.n = 0; blocks = 0; return MMDA_OK; }
      for (int i = L.n; i <= SPLITK_JOBS_MAX; ++i) L.start[i] = blocks;
      for (int i = L.n; i < SPLITK_JOBS_MAX; ++i) L.j[i] = L.j[0];
      if (form == 1) hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3(blocks), dim3(256), 0, s, L);
      else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, L);
      MMDA_CHECK_LAUNCH("mmda_splitk_reduce");
      L.n = 0; blocks = 0;
      return MMDA_OK;
    };
    for (int i = 0; i < n; ++i) {
      const SplitKJob& J = jobs[i];
      if ((is_vec(J) ? 1 : 0) != form) continue;
      if (L.n == SPLITK_JOBS_MAX) { const int rc = flush(); if (rc) return rc; }
      L.j[L.n] = J;
      L.start[L.n] = blocks;
      const int64_t elems = form == 1 ? (int64_t)J.M * (J.ldn >> 2) : (int64_t)J.M * (J.N + (J.bias_grad ? 1 : 0)) * J.batch;
      blocks += (int)((elems + 255) / 256);
      L.n++;
    }
    const int rc = flush();
    if (rc) return rc;
  }
  return MMDA_OK;
}
