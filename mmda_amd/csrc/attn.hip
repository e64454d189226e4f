// Self-attention core of the fusion layer (reference: nn.TransformerEncoderLayer(d_model=hs, nhead=2) applied to the
// SIX utterance tokens, models.py:160-161,243-245).  S = 6 regardless of sequence length, so the whole (sample, head)
// problem - q,k,v of 6 x hd, a 6x6 score matrix - lives in one 64-lane wavefront's LDS/registers; this is latency
// work, not a GEMM.  One workgroup (one wave) per (sample, head).
#include "common.h"
#include "rowlocal.h"

namespace {

constexpr int MAXS = 8;

__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ qkv, int S, int B, int E, int nhead,
                                                      float* ctx, float* probs, float p, uint64_t seed, int site) {
  extern __shared__ float sm[];
  const int hd = E / nhead;
  const int b = blockIdx.x / nhead, h = blockIdx.x % nhead;
  float* q = sm; float* k = q + S * hd; float* v = k + S * hd; float* pr = v + S * hd;   // pr: S*S
  const int tid = threadIdx.x;
  for (int i = tid; i < S * hd; i += 64) {
    int s = i / hd, d = i % hd;
    const float* row = qkv + ((int64_t)s * B + b) * 3 * E + h * hd + d;
    q[i] = row[0]; k[i] = row[E]; v[i] = row[2 * E];
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)hd);
  for (int e = tid; e < S * S; e += 64) {
    int i = e / S, j = e % S;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc += q[i * hd + d] * k[j * hd + d];
    pr[e] = acc * scale;
  }
  __syncthreads();
  if (tid < S) {
    float m = -INFINITY;
    for (int j = 0; j < S; ++j) m = fmaxf(m, pr[tid * S + j]);
    float sum = 0.f;
    for (int j = 0; j < S; ++j) { float e = expf(pr[tid * S + j] - m); pr[tid * S + j] = e; sum += e; }
    float inv = 1.0f / sum;
    for (int j = 0; j < S; ++j) {
      float pv = pr[tid * S + j] * inv;
      int64_t pi = ((int64_t)blockIdx.x * S + tid) * S + j;
      probs[pi] = pv;                                               // stash the un-dropped probabilities
      pr[tid * S + j] = pv * drop_mul(p, seed, site, (uint64_t)pi);
    }
  }
  __syncthreads();
  for (int i = tid; i < S * hd; i += 64) {
    int s = i / hd, d = i % hd;
    float acc = 0.f;
    for (int j = 0; j < S; ++j) acc += pr[s * S + j] * v[j * hd + d];
    ctx[((int64_t)s * B + b) * E + h * hd + d] = acc;
  }
}

__global__ __launch_bounds__(64) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                      const float* __restrict__ dctx, int S, int B, int E, int nhead,
                                                      float* dqkv, float p, uint64_t seed, int site) {
  extern __shared__ float sm[];
  const int hd = E / nhead;
  const int b = blockIdx.x / nhead, h = blockIdx.x % nhead;
  float* q = sm; float* k = q + S * hd; float* v = k + S * hd; float* dc = v + S * hd;
  float* P = dc + S * hd;        // S*S probabilities
  float* Pd = P + S * S;         // dropped probabilities
  float* dS = Pd + S * S;        // score gradients
  const int tid = threadIdx.x;
  for (int i = tid; i < S * hd; i += 64) {
    int s = i / hd, d = i % hd;
    const float* row = qkv + ((int64_t)s * B + b) * 3 * E + h * hd + d;
    q[i] = row[0]; k[i] = row[E]; v[i] = row[2 * E];
    dc[i] = dctx[((int64_t)s * B + b) * E + h * hd + d];
  }
  for (int e = tid; e < S * S; e += 64) {
    int64_t pi = (int64_t)blockIdx.x * S * S + e;
    float pv = probs[pi];
    P[e] = pv;
    Pd[e] = pv * drop_mul(p, seed, site, (uint64_t)pi);
  }
  __syncthreads();
  // dP[i][j] = (dctx[i] . v[j]) * dropmul  (kept in dS for now)
  for (int e = tid; e < S * S; e += 64) {
    int i = e / S, j = e % S;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc += dc[i * hd + d] * v[j * hd + d];
    dS[e] = acc * drop_mul(p, seed, site, (uint64_t)((int64_t)blockIdx.x * S * S + e));
  }
  __syncthreads();
  if (tid < S) {
    float dot = 0.f;
    for (int j = 0; j < S; ++j) dot += dS[tid * S + j] * P[tid * S + j];
    for (int j = 0; j < S; ++j) dS[tid * S + j] = P[tid * S + j] * (dS[tid * S + j] - dot);
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)hd);
  for (int i = tid; i < S * hd; i += 64) {
    int s = i / hd, d = i % hd;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < S; ++j) {
      dq += dS[s * S + j] * k[j * hd + d];
      dk += dS[j * S + s] * q[j * hd + d];
      dv += Pd[j * S + s] * dc[j * hd + d];
    }
    float* row = dqkv + ((int64_t)s * B + b) * 3 * E + h * hd + d;
    row[0] = dq * scale; row[E] = dk * scale; row[2 * E] = dv;
  }
}

// ------------------------------------------------------------------------------------------------ head width 64, S = 6
// The production shape (hidden 128, two heads, six tokens): the head width is the wave width, so lane d keeps column d of every
// q, k and v row in registers (18 values), a score is one wave reduction, and after the reductions every lane holds the whole
// 6 x 6 matrix -- softmax, dropout and the context product need no LDS and no barrier.  Same arithmetic order per element and
// the same dropout indices as the generic kernels above.

__global__ __launch_bounds__(64) void attn_fwd_hd64_kernel(const float* __restrict__ qkv, int B, float* ctx, float* probs, float p,
                                                           uint64_t seed, int site, int nhead) {
  attn_fwd_hd64_one(qkv, B, ctx, probs, p, seed, site, nhead, (int)blockIdx.x, (int)threadIdx.x);      // (rowlocal.h)
}

__global__ __launch_bounds__(64) void attn_bwd_hd64_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                           const float* __restrict__ dctx, int B, float* dqkv, float p, uint64_t seed,
                                                           int site, int nhead) {
  attn_bwd_hd64_one(qkv, probs, dctx, B, dqkv, p, seed, site, nhead, (int)blockIdx.x, (int)threadIdx.x);      // (rowlocal.h)
}

}  // namespace

extern "C" int mmda_attn_fwd(const float* qkv, int S, int B, int E, int nhead, float* ctx, float* probs,
                             float drop_p, uint64_t seed, int site, void* stream) {
  if (!qkv || !ctx || !probs || S <= 0 || S > MAXS || B <= 0 || nhead <= 0 || E % nhead) return MMDA_EINVAL;
  int hd = E / nhead;
  if (hd == 64 && S == S6K) {
    hipLaunchKernelGGL(attn_fwd_hd64_kernel, dim3(B * nhead), dim3(64), 0, (hipStream_t)stream, qkv, B, ctx, probs, drop_p, seed, site,
                       nhead);
    MMDA_CHECK_LAUNCH("mmda_attn_fwd");
    return MMDA_OK;
  }
  size_t lds = sizeof(float) * (3 * S * hd + S * S);
  if (lds > 64 * 1024) return MMDA_EINVAL;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * nhead), dim3(64), lds, (hipStream_t)stream, qkv, S, B, E, nhead, ctx, probs,
                     drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_attn_fwd");
  return MMDA_OK;
}

extern "C" int mmda_attn_bwd(const float* qkv, const float* probs, const float* dctx, int S, int B, int E, int nhead,
                             float* dqkv, float drop_p, uint64_t seed, int site, void* stream) {
  if (!qkv || !probs || !dctx || !dqkv || S <= 0 || S > MAXS || B <= 0 || nhead <= 0 || E % nhead) return MMDA_EINVAL;
  int hd = E / nhead;
  if (hd == 64 && S == S6K) {
    hipLaunchKernelGGL(attn_bwd_hd64_kernel, dim3(B * nhead), dim3(64), 0, (hipStream_t)stream, qkv, probs, dctx, B, dqkv, drop_p, seed,
                       site, nhead);
    MMDA_CHECK_LAUNCH("mmda_attn_bwd");
    return MMDA_OK;
  }
  size_t lds = sizeof(float) * (4 * S * hd + 3 * S * S);
  if (lds > 64 * 1024) return MMDA_EINVAL;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * nhead), dim3(64), lds, (hipStream_t)stream, qkv, probs, dctx, S, B, E, nhead,
                     dqkv, drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_attn_bwd");
  return MMDA_OK;
}
