// Row-wise kernels: LayerNorm forward/backward (wave-per-row shuffle reductions), embedding gather / scatter-add,
// small elementwise helpers.  All HBM-bound; loads are lane-consecutive (coalesced 256 B per wave instruction).
#include "common.h"

namespace {

constexpr int LN_MAXQ = 16;   // up to 16*64 = 1024 columns cached in registers per lane

__device__ __forceinline__ int64_t perm_row(int row, int S, int Bp) {
  // (s,b) row -> (b,s) row when a permutation is requested
  if (S <= 0) return row;
  int s = row / Bp, b = row % Bp;
  return (int64_t)b * S + s;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(mmda_ln_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.rows) return;
  const int n = a.n;
  float v[LN_MAXQ];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXQ; ++q) {
    int i = lane + 64 * q;
    float x = 0.f;
    if (i < n) {
      int64_t idx = (int64_t)row * n + i;
      x = act_fwd(a.act, a.x[idx]);
      if (a.res) x += a.res[idx] * drop_mul(a.drop_p, a.drop_seed, a.drop_site, (uint64_t)idx);
    }
    v[q] = x;
    s += x;
  }
  const float mean = wave_sum(s) / n;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXQ; ++q) {
    int i = lane + 64 * q;
    float d = (i < n) ? v[q] - mean : 0.f;
    ss += d * d;
  }
  const float var = wave_sum(ss) / n;
  const float rstd = 1.0f / sqrtf(var + a.eps);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
  const int64_t orow = perm_row(row, a.permute_S, a.permute_B);
#pragma unroll
  for (int q = 0; q < LN_MAXQ; ++q) {
    int i = lane + 64 * q;
    if (i < n) a.y[orow * n + i] = (v[q] - mean) * rstd * a.gamma[i] + a.beta[i];
  }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(mmda_ln_bwd_args a) {
  __shared__ float red[2][4][LN_MAXQ * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.n;
  float dg[LN_MAXQ], db[LN_MAXQ];
#pragma unroll
  for (int q = 0; q < LN_MAXQ; ++q) { dg[q] = 0.f; db[q] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < a.rows; row += gridDim.x * 4) {
    const float mean = a.mean[row], rstd = a.rstd[row];
    const int64_t drow = perm_row(row, a.permute_S, a.permute_B);
    float xh[LN_MAXQ], gdy[LN_MAXQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
      int i = lane + 64 * q;
      xh[q] = 0.f; gdy[q] = 0.f;
      if (i < n) {
        int64_t idx = (int64_t)row * n + i;
        float x = act_fwd(a.act, a.x[idx]);
        if (a.res) x += a.res[idx] * drop_mul(a.drop_p, a.drop_seed, a.drop_site, (uint64_t)idx);
        float dy = a.dy[drow * n + i];
        xh[q] = (x - mean) * rstd;
        gdy[q] = dy * a.gamma[i];
        dg[q] += dy * xh[q];
        db[q] += dy;
        s1 += gdy[q];
        s2 += gdy[q] * xh[q];
      }
    }
    s1 = wave_sum(s1) / n;
    s2 = wave_sum(s2) / n;
#pragma unroll
    for (int q = 0; q < LN_MAXQ; ++q) {
      int i = lane + 64 * q;
      if (i < n) {
        int64_t idx = (int64_t)row * n + i;
        float dxp = rstd * (gdy[q] - s1 - xh[q] * s2);
        if (a.d_x) {
          float d = dxp * act_bwd(a.act, a.x[idx]);
          a.d_x[idx] = a.accumulate_dx ? a.d_x[idx] + d : d;
        }
        if (a.d_res) a.d_res[idx] = dxp * drop_mul(a.drop_p, a.drop_seed, a.drop_site, (uint64_t)idx);
      }
    }
  }
  // reduce the per-wave column partials across the block's 4 waves, then one atomic per column per block
  const int nq = (n + 63) / 64;
#pragma unroll
  for (int q = 0; q < LN_MAXQ; ++q) {
    if (q < nq) { red[0][wave][q * 64 + lane] = dg[q]; red[1][wave][q * 64 + lane] = db[q]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float g = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
    float b = red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i];
    if (a.dgamma) atomicAdd(&a.dgamma[i], g);
    if (a.dbeta) atomicAdd(&a.dbeta[i], b);
  }
}

__global__ void embed_gather_kernel(const float* __restrict__ W, const int64_t* __restrict__ ids, int rows, int dim, float* out) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = W + ids[row] * (int64_t)dim;
  float* dst = out + (int64_t)row * dim;
  for (int i = threadIdx.x & 63; i < dim; i += 64) dst[i] = src[i];
}

__global__ void embed_scatter_kernel(float* dW, const int64_t* __restrict__ ids, int rows, int dim, const float* __restrict__ dX) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* dst = dW + ids[row] * (int64_t)dim;
  const float* src = dX + (int64_t)row * dim;
  for (int i = threadIdx.x & 63; i < dim; i += 64) atomicAdd(&dst[i], src[i]);   // 256-B contiguous per wave-instruction
}

__global__ void add_kernel(const float* a, const float* b, float* y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}
__global__ void sigmoid_bwd_kernel(float* d, const float* y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = y[i];
    d[i] *= s * (1.f - s);
  }
}

__global__ void act_drop_fwd_kernel(const float* z, float* h, int64_t n, int act, float p, uint64_t seed, int site) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    h[i] = act_fwd(act, z[i]) * drop_mul(p, seed, site, (uint64_t)i);
}
__global__ void act_drop_bwd_kernel(const float* dh, const float* z, float* dz, int64_t n, int act, float p, uint64_t seed, int site) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dz[i] = dh[i] * drop_mul(p, seed, site, (uint64_t)i) * act_bwd(act, z[i]);
}

int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mmda_layernorm_fwd(const mmda_ln_args* a, void* stream) {
  if (!a || !a->x || !a->y || !a->gamma || !a->beta || a->rows < 0 || a->n <= 0 || a->n > LN_MAXQ * 64) return MMDA_EINVAL;
  if (a->permute_S > 0 && (a->permute_B <= 0 || a->permute_S * a->permute_B != a->rows)) return MMDA_EINVAL;
  if (a->rows == 0) return MMDA_OK;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(a->rows, 4)), dim3(256), 0, (hipStream_t)stream, *a);
  MMDA_CHECK_LAUNCH("mmda_layernorm_fwd");
  return MMDA_OK;
}

extern "C" int mmda_layernorm_bwd(const mmda_ln_bwd_args* a, void* stream) {
  if (!a || !a->dy || !a->x || !a->gamma || !a->mean || !a->rstd || a->rows < 0 || a->n <= 0 || a->n > LN_MAXQ * 64) return MMDA_EINVAL;
  if (a->permute_S > 0 && (a->permute_B <= 0 || a->permute_S * a->permute_B != a->rows)) return MMDA_EINVAL;
  if (a->rows == 0) return MMDA_OK;
  int blocks = ceil_div(a->rows, 4 * 2);          // 2 rows per wave: enough blocks to fill the chip at rows ~ T*B = 1600
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
  MMDA_CHECK_LAUNCH("mmda_layernorm_bwd");
  return MMDA_OK;
}

extern "C" int mmda_embed_gather(const float* W, const int64_t* ids, int rows, int dim, float* out, void* stream) {
  if (!W || !ids || !out || rows < 0 || dim <= 0) return MMDA_EINVAL;
  if (rows == 0) return MMDA_OK;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, W, ids, rows, dim, out);
  MMDA_CHECK_LAUNCH("mmda_embed_gather");
  return MMDA_OK;
}

extern "C" int mmda_embed_scatter_add(float* dW, const int64_t* ids, int rows, int dim, const float* dX, void* stream) {
  if (!dW || !ids || !dX || rows < 0 || dim <= 0) return MMDA_EINVAL;
  if (rows == 0) return MMDA_OK;
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, dW, ids, rows, dim, dX);
  MMDA_CHECK_LAUNCH("mmda_embed_scatter_add");
  return MMDA_OK;
}

extern "C" int mmda_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  if (!a || !b || !y || n < 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
  MMDA_CHECK_LAUNCH("mmda_add");
  return MMDA_OK;
}

extern "C" int mmda_sigmoid_bwd_inplace(float* d, const float* y, int64_t n, void* stream) {
  if (!d || !y || n < 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, d, y, n);
  MMDA_CHECK_LAUNCH("mmda_sigmoid_bwd_inplace");
  return MMDA_OK;
}

extern "C" int mmda_act_dropout_fwd(const float* z, float* h, int64_t n, int act, float drop_p, uint64_t seed, int site, void* stream) {
  if (!z || !h || n < 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  hipLaunchKernelGGL(act_drop_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, z, h, n, act, drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_act_dropout_fwd");
  return MMDA_OK;
}

extern "C" int mmda_act_dropout_bwd(const float* dh, const float* z, float* dz, int64_t n, int act, float drop_p, uint64_t seed,
                                    int site, void* stream) {
  if (!dh || !z || !dz || n < 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  hipLaunchKernelGGL(act_drop_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dh, z, dz, n, act, drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_act_dropout_bwd");
  return MMDA_OK;
}
