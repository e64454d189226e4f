// Row-local pieces of the fusion block's backward pass as device functions: one LayerNorm-backward row (wave per row) and the
// attention backward of one (sample, head) at head width 64 and six tokens (wave per pair).  Shared by the stand-alone kernels
// (norm.hip, attn.hip) and the fused per-sample kernels (fused_rows.hip) -- one body, the same arithmetic order and dropout indices.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ int64_t perm_row(int row, int S, int Bp) {
  // (s,b) row -> (b,s) row when a permutation is requested
  if (S <= 0) return row;
  int s = row / Bp, b = row % Bp;
  return (int64_t)b * S + s;
}

// One row of the LayerNorm backward (y = LN(act(x) + res * dropmask) * gamma + beta): d_x / d_res of the row, and the row's terms of
// dgamma / dbeta added into the caller's per-lane column partials (dg, db).  Called by all 64 lanes of a wave.
template <int NQ>
__device__ __forceinline__ void ln_bwd_row(const mmda_ln_bwd_args& a, int row, int lane, float (&dg)[NQ], float (&db)[NQ]) {
  const int n = a.n;
  const float mean = a.mean[row], rstd = a.rstd[row];
  const int64_t drow = perm_row(row, a.permute_S, a.permute_B);
  float xh[NQ], gdy[NQ];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int i = lane + 64 * q;
    xh[q] = 0.f; gdy[q] = 0.f;
    if (i < n) {
      int64_t idx = (int64_t)row * n + i;
      float x = act_fwd_p(a.act, a.x[idx], a.actp, (uint64_t)idx);
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
  float dslope = 0.f;                                  // PReLU: d(slope) = sum of dx_pre * z over the elements with z <= 0
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int i = lane + 64 * q;
    if (i < n) {
      int64_t idx = (int64_t)row * n + i;
      float dxp = rstd * (gdy[q] - s1 - xh[q] * s2);
      if (a.d_x) {
        const float z = a.x[idx];
        float d = dxp * act_bwd_p(a.act, z, a.actp, (uint64_t)idx);
        a.d_x[idx] = a.accumulate_dx ? a.d_x[idx] + d : d;
        if (a.act == MMDA_ACT_PRELU && z <= 0.f) dslope += dxp * z;
      }
      if (a.d_res) a.d_res[idx] = dxp * drop_mul(a.drop_p, a.drop_seed, a.drop_site, (uint64_t)idx);
    }
  }
  if (a.act == MMDA_ACT_PRELU && a.d_x && a.actp.dslope) {       // problem-uniform
    dslope = wave_sum(dslope);
    if (lane == 0) atomicAdd(a.actp.dslope, dslope);
  }
}

// One row of the LayerNorm forward (wave per row): y = LN(act(x) + res * dropmask) * gamma + beta, mean / rstd stored, optional bf16
// copy of y (see mmda_ln_args).
template <int NQ>
__device__ __forceinline__ void ln_fwd_row(const mmda_ln_args& a, int row, int lane) {
  const int n = a.n;
  float v[NQ];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int i = lane + 64 * q;
    float x = 0.f;
    if (i < n) {
      int64_t idx = (int64_t)row * n + i;
      x = act_fwd_p(a.act, a.x[idx], a.actp, (uint64_t)idx);
      if (a.res) x += a.res[idx] * drop_mul(a.drop_p, a.drop_seed, a.drop_site, (uint64_t)idx);
    }
    v[q] = x;
    s += x;
  }
  const float mean = wave_sum(s) / n;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
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
  for (int q = 0; q < NQ; ++q) {
    int i = lane + 64 * q;
    float y = 0.f;
    if (i < n) { y = (v[q] - mean) * rstd * a.gamma[i] + a.beta[i]; if (a.y) a.y[orow * n + i] = y; }
    if (a.y_bf16 && i < a.ld_bf16) reinterpret_cast<unsigned short*>(a.y_bf16)[orow * a.ld_bf16 + i] = f2bf(y);      // (zero in the padding)
  }
}

constexpr int S6K = 6;

// attention forward of pair bh = b * nhead + h at head width 64, six tokens: lane d keeps column d of every q, k and v row (18 values),
// a score is one wave reduction, every lane then holds the whole 6 x 6 matrix -- softmax, dropout and the context product need no LDS
__device__ __forceinline__ void attn_fwd_hd64_one(const float* __restrict__ qkv, int B, float* ctx, float* probs, float p, uint64_t seed,
                                                  int site, int nhead, int bh, int d) {
  constexpr int S = S6K, hd = 64;
  const int E = hd * nhead;
  const int b = bh / nhead, h = bh % nhead;
  float q[S], k[S], v[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float* row = qkv + ((int64_t)s * B + b) * 3 * E + h * hd + d;
    q[s] = row[0]; k[s] = row[E]; v[s] = row[2 * E];
  }
  const float scale = 1.0f / sqrtf((float)hd);
  float pr[S][S];
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) pr[i][j] = wave_sum(q[i] * k[j]) * scale;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < S; ++j) m = fmaxf(m, pr[i][j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < S; ++j) { pr[i][j] = expf(pr[i][j] - m); sum += pr[i][j]; }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const float pv = pr[i][j] * inv;
      const int64_t pi = ((int64_t)bh * S + i) * S + j;
      if (d == 0) probs[pi] = pv;                                   // every lane holds the same value
      pr[i][j] = pv * drop_mul(p, seed, site, (uint64_t)pi);
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < S; ++j) acc += pr[s][j] * v[j];
    ctx[((int64_t)s * B + b) * E + h * hd + d] = acc;
  }
}


// attention backward of pair bh = b * nhead + h (head width 64 = the wave: lane d owns column d of every q, k, v, d_ctx row)
__device__ __forceinline__ void attn_bwd_hd64_one(const float* __restrict__ qkv, const float* __restrict__ probs, const float* __restrict__ dctx,
                                                  int B, float* dqkv, float p, uint64_t seed, int site, int nhead, int bh, int d) {
  constexpr int S = S6K, hd = 64;
  const int E = hd * nhead;
  const int b = bh / nhead, h = bh % nhead;
  float q[S], k[S], v[S], dc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float* row = qkv + ((int64_t)s * B + b) * 3 * E + h * hd + d;
    q[s] = row[0]; k[s] = row[E]; v[s] = row[2 * E];
    dc[s] = dctx[((int64_t)s * B + b) * E + h * hd + d];
  }
  float P[S][S], Pd[S][S], dS[S][S];
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int64_t pi = (int64_t)bh * S * S + i * S + j;
      const float mul = drop_mul(p, seed, site, (uint64_t)pi);
      P[i][j] = probs[pi];
      Pd[i][j] = P[i][j] * mul;
      dS[i][j] = wave_sum(dc[i] * v[j]) * mul;                      // dP
    }
#pragma unroll
  for (int i = 0; i < S; ++i) {
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < S; ++j) dot += dS[i][j] * P[i][j];
#pragma unroll
    for (int j = 0; j < S; ++j) dS[i][j] = P[i][j] * (dS[i][j] - dot);
  }
  const float scale = 1.0f / sqrtf((float)hd);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    float dq = 0.f, dk = 0.f, dv = 0.f;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      dq += dS[s][j] * k[j];
      dk += dS[j][s] * q[j];
      dv += Pd[j][s] * dc[j];
    }
    float* row = dqkv + ((int64_t)s * B + b) * 3 * E + h * hd + d;
    row[0] = dq * scale; row[E] = dk * scale; row[2 * E] = dv;
  }
}

}  // namespace
