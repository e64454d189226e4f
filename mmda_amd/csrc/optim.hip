// clip_grad_value_ + Adam fused over one flat fp32 bucket (reference solver.py:185-186 and :97-99: torch.optim.Adam with
// lr only -> betas (0.9, 0.999), eps 1e-8, no weight decay; --weight_decay is parsed but never used).  Pure HBM stream:
// reads p,g,m,v and writes p,m,v = 28 B per parameter, float4 per lane.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float clip,
                                      float gscale, float step_size, float inv_bc2_sqrt) {
  // every rounding spelled out: the dense kernel and the per-row kernel below must give the same bits (the fused step runs part of the
  // bucket through each), whatever the compiler would contract in either loop
  g = __fmul_rn(g, gscale);
  g = fminf(fmaxf(g, -clip), clip);
  m = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, g));
  v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, g), g));
  const float denom = __fmaf_rn(sqrtf(v), inv_bc2_sqrt, eps);
  p = __fmaf_rn(-step_size, __fdiv_rn(m, denom), p);
}

// wait_flag != nullptr (flag join, common.h): workgroup 0 does not finish before that word reaches wait_value -- so the completion of this
// launch on its stream implies the completion of the other stream's chain (the update itself does not depend on it).
__global__ __launch_bounds__(256) void clamp_adam_kernel(float* p, const float* __restrict__ g, float* m, float* v, int64_t n,
                                                         float b1, float b2, float eps, float clip, float gscale,
                                                         float step_size, float inv_bc2_sqrt, const unsigned* wait_flag,
                                                         unsigned wait_value, unsigned* wait_err) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    adam1(pp.x, gg.x, mm.x, vv.x, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
    adam1(pp.y, gg.y, mm.y, vv.y, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
    adam1(pp.z, gg.z, mm.z, vv.z, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
    adam1(pp.w, gg.w, mm.w, vv.w, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    adam1(p[i], g[i], m[i], v[i], b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
  if (blockIdx.x == 0) flag_wait(wait_flag, wait_value, wait_err);
}

// The same update over the ROWS of a (rows, dim) table whose mask byte equals `want` (dim % 4 == 0 or not: scalar tail per row).
// The embedding matrix is 6 of the model's 10.8 M parameters and a step touches at most T*B of its V rows: the rows a batch does not
// touch have a zero gradient that is known before the backward pass ends, so their update runs early, beside the last recurrence
// (mask 0), and only the touched rows wait for the scattered gradient (mask 1).  One wave per row, lanes along the row.
__global__ __launch_bounds__(256) void clamp_adam_rows_kernel(float* p, const float* __restrict__ g, float* m, float* v, int rows, int dim,
                                                              const unsigned char* __restrict__ mask, int want, float b1, float b2,
                                                              float eps, float clip, float gscale, float step_size, float inv_bc2_sqrt) {
  const int lane = threadIdx.x & 63;
  const int wpg = (int)gridDim.x * 4;
  for (int row = (int)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += wpg) {
    if ((int)mask[row] != want) continue;                // wave-uniform
    const int64_t o = (int64_t)row * dim;
    if ((dim & 3) == 0 && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0)) {      // launch-uniform
      float4* p4 = reinterpret_cast<float4*>(p + o);
      const float4* g4 = reinterpret_cast<const float4*>(g + o);
      float4* m4 = reinterpret_cast<float4*>(m + o);
      float4* v4 = reinterpret_cast<float4*>(v + o);
      for (int i = lane; i < (dim >> 2); i += 64) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
        adam1(pp.y, gg.y, mm.y, vv.y, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
        adam1(pp.z, gg.z, mm.z, vv.z, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
        adam1(pp.w, gg.w, mm.w, vv.w, b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
      }
    } else {
      for (int i = lane; i < dim; i += 64) adam1(p[o + i], g[o + i], m[o + i], v[o + i], b1, b2, eps, clip, gscale, step_size, inv_bc2_sqrt);
    }
  }
}

__global__ void mark_rows_kernel(unsigned char* mask, int rows, const int64_t* __restrict__ ids, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  if (id >= 0 && id < rows) mask[id] = 1;
}

// clip_grad_value_ + torch.optim.RMSprop (alpha, eps; no momentum, not centered, no weight decay: the reference constructs its
// optimizer as config.optimizer(params, lr=...), solver.py:97-99, so every other argument is torch's default)
__global__ __launch_bounds__(256) void clamp_rmsprop_kernel(float* p, const float* __restrict__ g, float* sq, int64_t n, float lr,
                                                            float alpha, float eps, float clip, float gscale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gg = fminf(fmaxf(g[i] * gscale, -clip), clip);
    const float s = alpha * sq[i] + (1.0f - alpha) * gg * gg;
    sq[i] = s;
    p[i] -= lr * gg / (sqrtf(s) + eps);
  }
}

// two buffers cleared by one launch (16-byte stores; counts in floats, multiples of 4, 16-byte aligned bases)
__global__ __launch_bounds__(256) void zero2_kernel(float4* a, int64_t na4, float4* b, int64_t nb4) {
  const float4 z = {0.f, 0.f, 0.f, 0.f};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < na4 + nb4; i += stride) {
    if (i < na4) a[i] = z;
    else b[i - na4] = z;
  }
}

__global__ void clamp_kernel(float* g, int64_t n, float clip) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    g[i] = fminf(fmaxf(g[i], -clip), clip);
}

}  // namespace

extern "C" int mmda_clamp_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float clip, float grad_scale, int step, void* stream) {
  return mmda_clamp_adam_wait(p, g, m, v, n, lr, beta1, beta2, eps, clip, grad_scale, step, nullptr, 0u, nullptr, stream);
}

// internal (misa.hip): the same launch, not complete before *wait_flag reaches wait_value (see clamp_adam_kernel)
int mmda_clamp_adam_wait(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float clip,
                         float grad_scale, int step, const unsigned* wait_flag, unsigned wait_value, unsigned* wait_err, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return MMDA_EINVAL;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MMDA_EINVAL;   // float4 path
  if (n == 0 && !wait_flag) return MMDA_OK;
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(clamp_adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps, clip,
                     grad_scale, step_size, inv_bc2_sqrt, wait_flag, wait_value, wait_err);
  MMDA_CHECK_LAUNCH("mmda_clamp_adam");
  return MMDA_OK;
}

// internal (misa.hip): a[0 .. na) = b[0 .. nb) = 0 in ONE launch (two memsets are two launches of ~5 us each on a latency-bound chain)
int mmda_zero2(float* a, int64_t na, float* b, int64_t nb, void* stream) {
  if (na < 0 || nb < 0 || (na && !a) || (nb && !b)) return MMDA_EINVAL;
  if (((uintptr_t)a | (uintptr_t)b) & 15 || (na & 3) || (nb & 3)) {           // odd shapes: the runtime's fills
    if (na && hipMemsetAsync(a, 0, sizeof(float) * na, (hipStream_t)stream) != hipSuccess) return MMDA_ELAUNCH;
    if (nb && hipMemsetAsync(b, 0, sizeof(float) * nb, (hipStream_t)stream) != hipSuccess) return MMDA_ELAUNCH;
    return MMDA_OK;
  }
  const int64_t n4 = (na + nb) >> 2;
  if (n4 == 0) return MMDA_OK;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(zero2_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(a), na >> 2,
                     reinterpret_cast<float4*>(b), nb >> 2);
  MMDA_CHECK_LAUNCH("mmda_zero2");
  return MMDA_OK;
}

extern "C" int mmda_mark_rows(unsigned char* mask, int rows, const int64_t* ids, int n, void* stream) {
  if (!mask || !ids || rows <= 0 || n < 0) return MMDA_EINVAL;
  if (hipMemsetAsync(mask, 0, (size_t)rows, (hipStream_t)stream) != hipSuccess) return MMDA_ELAUNCH;
  if (n == 0) return MMDA_OK;
  hipLaunchKernelGGL(mark_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mask, rows, ids, n);
  MMDA_CHECK_LAUNCH("mmda_mark_rows");
  return MMDA_OK;
}

extern "C" int mmda_clamp_adam_rows(float* p, const float* g, float* m, float* v, int rows, int dim, const unsigned char* mask, int want,
                                    float lr, float beta1, float beta2, float eps, float clip, float grad_scale, int step, void* stream) {
  if (!p || !g || !m || !v || !mask || rows < 0 || dim <= 0 || step < 1) return MMDA_EINVAL;
  if (rows == 0) return MMDA_OK;
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  int blocks = (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(clamp_adam_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, rows, dim, mask, want, beta1,
                     beta2, eps, clip, grad_scale, step_size, inv_bc2_sqrt);
  MMDA_CHECK_LAUNCH("mmda_clamp_adam_rows");
  return MMDA_OK;
}

extern "C" int mmda_clamp(float* g, int64_t n, float clip, void* stream) {
  if (!g || n < 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(clamp_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, n, clip);
  MMDA_CHECK_LAUNCH("mmda_clamp");
  return MMDA_OK;
}

extern "C" int mmda_clamp_rmsprop(float* p, const float* g, float* square_avg, int64_t n, float lr, float alpha, float eps, float clip,
                                  float grad_scale, void* stream) {
  if (!p || !g || !square_avg || n < 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(clamp_rmsprop_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, square_avg, n, lr, alpha, eps, clip,
                     grad_scale);
  MMDA_CHECK_LAUNCH("mmda_clamp_rmsprop");
  return MMDA_OK;
}
