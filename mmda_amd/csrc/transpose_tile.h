// fp32 transposes through 32 x 33 LDS tiles (device code shared by transpose_kernel in gemm_skinny.hip and by the merged launch at the
// start of a training step in lstm.hip) and the host-side launch table.
#pragma once
#include "common.h"

namespace {

constexpr int TR_MAX = 20;
struct TrLaunch { mmda_transpose_job j[TR_MAX]; int start[TR_MAX + 1]; int tx[TR_MAX]; int n; };

// block `b` of the launch table: one 32 x 32 tile, coalesced on both sides (256 threads)
__device__ __forceinline__ void transpose_block(const TrLaunch& L, int b, float (*tile)[33]) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < TR_MAX; ++k)
    if (k < L.n && b >= L.start[k]) pi = k;
  const mmda_transpose_job& J = L.j[pi];
  const int local = b - L.start[pi];
  const int bx = local % L.tx[pi], by = local / L.tx[pi];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = by * 32 + ty + 8 * i, c = bx * 32 + tx;
    tile[ty + 8 * i][tx] = (r < J.rows && c < J.cols) ? J.src[(int64_t)min(r, J.rows - 1) * J.ld + min(c, J.cols - 1)] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = bx * 32 + ty + 8 * i, r = by * 32 + tx;
    if (c < J.cols && r < J.rows) J.dst[(int64_t)c * J.ldd + r] = tile[tx][ty + 8 * i];
  }
}

// fills L from jobs[0 .. n), n <= TR_MAX (empty jobs are dropped); blocks = workgroups of 256 threads the table needs
inline int tr_build(const mmda_transpose_job* jobs, int n, TrLaunch& L, int& blocks) {
  L.n = 0;
  blocks = 0;
  if (n < 0 || n > TR_MAX || (n > 0 && !jobs)) return MMDA_EINVAL;
  for (int i = 0; i < n; ++i) {
    const mmda_transpose_job& j = jobs[i];
    if (!j.src || !j.dst || j.rows < 0 || j.cols < 0 || j.ld < j.cols || j.ldd < j.rows) return MMDA_EINVAL;
    if (j.rows == 0 || j.cols == 0) continue;
    const int k = L.n++;
    L.j[k] = j; L.tx[k] = (j.cols + 31) / 32; L.start[k] = blocks;
    blocks += L.tx[k] * ((j.rows + 31) / 32);
  }
  for (int k = L.n; k <= TR_MAX; ++k) L.start[k] = blocks;
  for (int k = L.n; k < TR_MAX; ++k) { L.j[k] = L.n ? L.j[0] : mmda_transpose_job{}; L.tx[k] = 1; }
  return MMDA_OK;
}

}  // namespace
