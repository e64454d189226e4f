// clip_grad_value_ + Adam fused over one flat fp32 bucket (reference solver.py:185-186 and :97-99: torch.optim.Adam with
// lr only -> betas (0.9, 0.999), eps 1e-8, no weight decay; --weight_decay is parsed but never used).  Pure HBM stream:
// reads p,g,m,v and writes p,m,v = 28 B per parameter, float4 per lane.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float clip,
                                      float gscale, float step_size, float inv_bc2_sqrt) {
  g *= gscale;
  g = fminf(fmaxf(g, -clip), clip);
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void clamp_adam_kernel(float* p, const float* __restrict__ g, float* m, float* v, int64_t n,
                                                         float b1, float b2, float eps, float clip, float gscale,
                                                         float step_size, float inv_bc2_sqrt) {
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

__global__ void clamp_kernel(float* g, int64_t n, float clip) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    g[i] = fminf(fmaxf(g[i], -clip), clip);
}

}  // namespace

extern "C" int mmda_clamp_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float clip, float grad_scale, int step, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return MMDA_EINVAL;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MMDA_EINVAL;   // float4 path
  if (n == 0) return MMDA_OK;
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(clamp_adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps, clip,
                     grad_scale, step_size, inv_bc2_sqrt);
  MMDA_CHECK_LAUNCH("mmda_clamp_adam");
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
