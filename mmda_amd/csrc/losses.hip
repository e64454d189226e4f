// Output heads and the loss functions of the reference solver (solver.py:373-462, utils/functions.py:49-109).
// Every loss kernel produces the value AND the gradient w.r.t. its inputs in the same pass ("gradient in forward"):
// the total loss is a fixed weighted sum (solver.py:175-181), so the backward seed of each term is its weight.
// These are batch-statistic reductions over (B, 128)-sized tensors: latency work on a handful of workgroups.
#include "common.h"
#include "splitk.h"

namespace {

// ------------------------------------------------------------------------------------------------ heads
__global__ void heads_fwd_kernel(const float* __restrict__ logits, int B, int ncls, float thr, float* tcp, float* scores,
                                 float* labels, float p, uint64_t seed, int site) {
  const int NC = 6 + ncls;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * NC; e += gridDim.x * blockDim.x) {
    int b = e / NC, c = e % NC;
    float z = logits[e];
    if (c < 6) {
      tcp[b * 6 + c] = sigmoidf_(z);
    } else {
      int k = c - 6;
      float s = sigmoidf_(z * drop_mul(p, seed, site, (uint64_t)(b * ncls + k)));
      scores[b * ncls + k] = s;
      labels[b * ncls + k] = s > thr ? 1.f : 0.f;
    }
  }
}

__global__ void heads_bwd_kernel(const float* __restrict__ tcp, const float* __restrict__ scores, const float* dtcp,
                                 const float* dscores, int B, int ncls, float* dlogits, float p, uint64_t seed, int site) {
  const int NC = 6 + ncls;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * NC; e += gridDim.x * blockDim.x) {
    int b = e / NC, c = e % NC;
    float g = 0.f;
    if (c < 6) {
      if (dtcp) { float t = tcp[b * 6 + c]; g = dtcp[b * 6 + c] * t * (1.f - t); }
    } else {
      int k = c - 6;
      if (dscores) {
        float s = scores[b * ncls + k];
        g = dscores[b * ncls + k] * s * (1.f - s) * drop_mul(p, seed, site, (uint64_t)(b * ncls + k));
      }
    }
    dlogits[e] = g;
  }
}

// A loss that several workgroups contribute to: every workgroup leaves its partial in parts[block] and ONE thread adds them in block
// order (float atomics added them in arrival order: the reported value changed in its last bits from run to run).
__global__ void loss_parts_finish_kernel(const float* __restrict__ parts, int n, float* loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float t = 0.f;
  for (int i = 0; i < n; ++i) t += parts[i];
  *loss += t;
}
__device__ __forceinline__ void loss_parts_add(const float* __restrict__ parts, int n, float* loss) {   // same, from inside a later kernel
  float t = 0.f;
  for (int i = 0; i < n; ++i) t += parts[i];
  *loss += t;
}

// ------------------------------------------------------------------------------------------------ cls (BCE)
// loss = sum_c (1/B) sum_b -(y log s + (1-y) log(1-s)), logs clamped at -100 (torch BCELoss);
// grad  = (s - y) / max(s (1-s), 1e-12) / B   (torch binary_cross_entropy_backward)
__global__ __launch_bounds__(256) void cls_kernel(const float* __restrict__ s, const float* __restrict__ y, int B, int ncls,
                                                  float scale, float* loss, float* ds) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int e = threadIdx.x; e < B * ncls; e += blockDim.x) {
    float sv = s[e], yv = y[e];
    float lp = fmaxf(logf(sv), -100.f), lq = fmaxf(log1pf(-sv), -100.f);
    acc += -(yv * lp + (1.f - yv) * lq);
    if (ds) ds[e] += scale * (sv - yv) / fmaxf(sv * (1.f - sv), 1e-12f) / B;
  }
  float t = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, t / B);
}

// ------------------------------------------------------------------------------------------------ conf (ConfidNet)
// One workgroup per class c (solver.py:458-460):
//   tcp  : mean_b (tcp - y*s)^2 / nnz
//   mcp  : CrossEntropyLoss on a 1-D input with float target = -sum_b y_b * log_softmax_over_batch(s)_b / nnz
__global__ __launch_bounds__(256) void conf_kernel(const float* __restrict__ s, const float* __restrict__ tcp,
                                                   const float* __restrict__ y, int B, int ncls, float scale, float* loss,
                                                   float* ds, float* dtcp) {
  __shared__ float red[16];
  __shared__ float smax[4];
  const int c = blockIdx.x;
  float nz = 0.f, sy = 0.f, mx = -INFINITY;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float yv = y[b * ncls + c];
    nz += (yv != 0.f) ? 1.f : 0.f;
    sy += yv;
    mx = fmaxf(mx, s[b * ncls + c]);
  }
  nz = block_sum(nz, red);
  sy = block_sum(sy, red);
  mx = wave_max(mx);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  float se = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) se += expf(s[b * ncls + c] - mx);
  se = block_sum(se, red);
  const float lse = mx + logf(se);
  float lt = 0.f, lm = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int e = b * ncls + c;
    float sv = s[e], yv = y[e], tv = tcp[e];
    float diff = tv - yv * sv;
    lt += diff * diff;
    lm += -yv * (sv - lse);
    float gt = 2.f * diff / (B * nz);
    if (dtcp) dtcp[e] += scale * gt;
    if (ds) ds[e] += scale * (-yv * gt + (-yv + sy * expf(sv - lse)) / nz);
  }
  lt = block_sum(lt, red);
  lm = block_sum(lm, red);
  if (threadIdx.x == 0 && loss) loss[blockIdx.x] = lt / (B * nz) + lm / nz;      // `loss` = this launch's partial array (one per class)
}

// ------------------------------------------------------------------------------------------------ cls + conf + recon + total
// The three small losses in ONE launch (block 0: cls, blocks 1..ncls: conf, the rest: recon) and, by the block that finishes
// last (device-scope ticket), the weighted total of all five (solver.py:175-181).  Launched after the diff and similarity
// losses on the same stream, so those sums are complete.  d_scores is shared by cls and conf: atomic adds here.
struct MiscLossArgs {
  const float* scores; const float* tcp; const float* emo; int B, ncls;
  float* d_scores; float* d_tcp;            // NULL: no gradients
  int conf_grads;                           // conf seeds gradients only with use_confidNet (solver.py:180-181)
  float conf_scale;
  const float* recon; const float* orig; int64_t n_recon; float recon_inv_n, recon_scale; float* d_recon; float* d_orig;
  float* L;                                 // cls, diff, sim, recon, conf, total, -, ticket
  float* parts;                             // one partial per workgroup (role order); the last workgroup adds them in that order
  float dw, sw, rw, cw; int use_conf, with_conf;
  int recon_blocks;
};

__global__ __launch_bounds__(256) void misc_losses_kernel(MiscLossArgs a) {
  __shared__ float red[16];
  __shared__ float smax[4];
  const int B = a.B, ncls = a.ncls;
  const int role = blockIdx.x;
  if (role == 0) {
    float acc = 0.f;
    for (int e = threadIdx.x; e < B * ncls; e += blockDim.x) {
      float sv = a.scores[e], yv = a.emo[e];
      float lp = fmaxf(logf(sv), -100.f), lq = fmaxf(log1pf(-sv), -100.f);
      acc += -(yv * lp + (1.f - yv) * lq);
      if (a.d_scores) atomicAdd(&a.d_scores[e], (sv - yv) / fmaxf(sv * (1.f - sv), 1e-12f) / B);
    }
    float t = block_sum(acc, red);
    if (threadIdx.x == 0) a.parts[role] = t / B;
  } else if (role <= (a.with_conf ? ncls : 0)) {
    const int c = role - 1;
    const float* s = a.scores; const float* y = a.emo;
    float nz = 0.f, sy = 0.f, mx = -INFINITY;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      float yv = y[b * ncls + c];
      nz += (yv != 0.f) ? 1.f : 0.f;
      sy += yv;
      mx = fmaxf(mx, s[b * ncls + c]);
    }
    nz = block_sum(nz, red);
    sy = block_sum(sy, red);
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    float se = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) se += expf(s[b * ncls + c] - mx);
    se = block_sum(se, red);
    const float lse = mx + logf(se);
    float lt = 0.f, lm = 0.f;
    const bool cg = a.conf_grads && a.d_scores;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      int e = b * ncls + c;
      float sv = s[e], yv = y[e], tv = a.tcp[e];
      float diff = tv - yv * sv;
      lt += diff * diff;
      lm += -yv * (sv - lse);
      float gt = 2.f * diff / (B * nz);
      if (cg) {
        atomicAdd(&a.d_tcp[e], a.conf_scale * gt);
        atomicAdd(&a.d_scores[e], a.conf_scale * (-yv * gt + (-yv + sy * expf(sv - lse)) / nz));
      }
    }
    lt = block_sum(lt, red);
    lm = block_sum(lm, red);
    if (threadIdx.x == 0) a.parts[role] = lt / (B * nz) + lm / nz;
  } else {
    const int rb = role - 1 - (a.with_conf ? ncls : 0);
    float acc = 0.f;
    for (int64_t e = rb * (int64_t)blockDim.x + threadIdx.x; e < a.n_recon; e += (int64_t)a.recon_blocks * blockDim.x) {
      float d = a.recon[e] - a.orig[e];
      acc += d * d;
      float g = 2.f * d * a.recon_inv_n * a.recon_scale;
      if (a.d_recon) a.d_recon[e] += g;
      if (a.d_orig) a.d_orig[e] -= g;
    }
    float t = block_sum(acc, red);
    if (threadIdx.x == 0) a.parts[role] = t * a.recon_inv_n;
  }
  // the last block to arrive sees every sum (release by the fence before the ticket, agent-scope reads after it)
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned* ticket = reinterpret_cast<unsigned*>(a.L + 7);
    const unsigned prev = atomicAdd(ticket, 1u);
    if (prev == gridDim.x - 1) {
      __threadfence();
      // cls, conf and recon: the workgroups' partials in role order (L[0], L[3], L[4] were cleared with the rest of the loss block)
      const int nconf = a.with_conf ? ncls : 0;
      // (agent-scope loads, sixteen in flight: one at a time is an L2 round trip each)
      auto ld = [&](int i) { return __hip_atomic_load(a.parts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      float cls = ld(0), conf = 0.f, rec = 0.f;
      for (int i = 0; i < nconf; ++i) conf += ld(1 + i);
      for (int i0 = 0; i0 < a.recon_blocks; i0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld(1 + nconf + min(i0 + u, a.recon_blocks - 1));      // (clamped: no load under a branch)
#pragma unroll
        for (int u = 0; u < 16; ++u) rec += (i0 + u < a.recon_blocks) ? v[u] : 0.f;
      }
      a.L[0] += cls; a.L[4] += conf; a.L[3] += rec;
      __threadfence();
      float v[5];
      for (int i = 0; i < 5; ++i) v[i] = __hip_atomic_load(a.L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float t = v[0] + a.dw * v[1] + a.sw * v[2] + a.rw * v[3];
      if (a.use_conf) t += a.cw * v[4];
      a.L[5] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ diff
// (1) per tensor: centre over the batch, divide rows by (detached L2 norm + 1e-6)            -> Ahat, invn
// (2) K_k = Ahat_k Ahat_k^T (B x B) for the six tensors (batched MFMA GEMM)
// (3) loss_pair(i,j) = mean((Ahat_i^T Ahat_j)^2) = sum(K_i o K_j)/D^2 ;  Ksum_i = sum_{j in pairs(i)} K_j * 2*scale/D^2
// (4) dAhat_i = Ksum_i Ahat_i (batched GEMM)
// (5) dx_i += dAhat_i*invn - colmean(dAhat_i*invn)   (norm detached, centring backward)
__global__ __launch_bounds__(256) void diff_prep_kernel(const float* __restrict__ x, int64_t stride, int B, int D, float* Ahat,
                                                        float* invn, float* mean) {
  const int k = blockIdx.x;
  const float* X = x + k * stride;
  float* M = mean + k * D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < B; ++r) {
      float v = X[(int64_t)r * D + i];
      v = (v != v) ? 0.f : fminf(fmaxf(v, -3.402823466e38f), 3.402823466e38f);   // nan_to_num
      s += v;
    }
    M[i] = s / B;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < B; r += 4) {
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) {
      float v = X[(int64_t)r * D + i];
      v = (v != v) ? 0.f : fminf(fmaxf(v, -3.402823466e38f), 3.402823466e38f);
      float c = v - M[i];
      ss += c * c;
    }
    ss = wave_sum(ss);
    float inv = 1.0f / (sqrtf(ss) + 1e-6f);
    if (lane == 0) invn[k * B + r] = inv;
    for (int i = lane; i < D; i += 64) {
      float v = X[(int64_t)r * D + i];
      v = (v != v) ? 0.f : fminf(fmaxf(v, -3.402823466e38f), 3.402823466e38f);
      Ahat[((int64_t)k * B + r) * D + i] = (v - M[i]) * inv;
    }
  }
}

// Parallel forms of diff_prep_kernel / diff_finish_kernel for the training shapes (D <= 128, B <= 64): one 1024-thread block per
// tensor = 128 columns x 8 row groups, the tensor's elements stay in registers; the serial per-column row loops of the
// generic kernels (one dependent load per row) cost 12 + 18 us of the step for 100 KB of data.
template <int RPT>
__global__ __launch_bounds__(1024) void diff_prep_fast_kernel(const float* __restrict__ x, int64_t stride, int B, int D, float* Ahat,
                                                              float* invn, float* mean) {
  __shared__ float P[8][128];
  __shared__ float R2[RPT * 8][2];
  const int k = blockIdx.x, tid = threadIdx.x, c = tid & 127, rg = tid >> 7;
  const int cc = min(c, D - 1);
  const bool c_ok = c < D;
  const float* X = x + k * stride;
  float v[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = rg + 8 * j;
    float val = X[(int64_t)min(r, B - 1) * D + cc];
    val = (val != val) ? 0.f : fminf(fmaxf(val, -3.402823466e38f), 3.402823466e38f);   // nan_to_num
    v[j] = (c_ok && r < B) ? val : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < RPT; ++j) s += v[j];
  P[rg][c] = s;
  __syncthreads();
  float m = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) m += P[g][c];
  m /= B;
  if (rg == 0 && c_ok) mean[k * D + c] = m;
  // row norms of the centred rows: a row's 128 columns sit in two waves
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const bool ok = c_ok && rg + 8 * j < B;
    v[j] = ok ? v[j] - m : 0.f;
    const float q = wave_sum(v[j] * v[j]);
    if ((tid & 63) == 0) R2[j * 8 + rg][c >> 6] = q;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = rg + 8 * j;
    const float inv = 1.0f / (sqrtf(R2[j * 8 + rg][0] + R2[j * 8 + rg][1]) + 1e-6f);
    if (c == 0 && r < B) invn[k * B + r] = inv;
    if (c_ok && r < B) Ahat[((int64_t)k * B + r) * D + c] = v[j] * inv;
  }
}

template <int RPT>
__global__ __launch_bounds__(1024) void diff_finish_fast_kernel(const float* __restrict__ dA, const float* __restrict__ invn, int B,
                                                                int D, int64_t stride, float* dx, const float* lparts, int nparts, float* loss) {
  __shared__ float P[8][128];
  const int k = blockIdx.x, tid = threadIdx.x, c = tid & 127, rg = tid >> 7;
  if (k == 0 && tid == 0 && loss) loss_parts_add(lparts, nparts, loss);        // the combine launch's loss partials, in block order
  const int cc = min(c, D - 1);
  const bool c_ok = c < D;
  float v[RPT], old[RPT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = rg + 8 * j, rc = min(r, B - 1);
    const float val = dA[((int64_t)k * B + rc) * D + cc] * invn[k * B + rc];
    old[j] = dx[k * stride + (int64_t)rc * D + cc];
    v[j] = (c_ok && r < B) ? val : 0.f;
    s += v[j];
  }
  P[rg][c] = s;
  __syncthreads();
  float cm = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) cm += P[g][c];
  cm /= B;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = rg + 8 * j;
    if (c_ok && r < B) dx[k * stride + (int64_t)r * D + c] = old[j] + v[j] - cm;
  }
}

struct PairList { int nt, np; int a[6], b[6]; };

__device__ __forceinline__ float sel6(const float (&v)[6], int i) {
  // register-resident select (a runtime-indexed local array would go to scratch)
  float r = v[0];
  r = i == 1 ? v[1] : r; r = i == 2 ? v[2] : r; r = i == 3 ? v[3] : r; r = i == 4 ? v[4] : r; r = i == 5 ? v[5] : r;
  return r;
}

__global__ __launch_bounds__(256) void diff_combine_kernel(const float* __restrict__ K, int B, int D, float scale, float* loss,
                                                           float* Ksum, PairList pl) {
  __shared__ float red[16];
  const int64_t BB = (int64_t)B * B;
  float acc = 0.f;
  const float gscale = 2.f * scale / ((float)D * D);
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < BB; e += (int64_t)gridDim.x * blockDim.x) {
    float kv[6], ks[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 6; ++t) kv[t] = t < pl.nt ? K[t * BB + e] : 0.f;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      if (p < pl.np) {
        const int i = pl.a[p], j = pl.b[p];
        const float ki = sel6(kv, i), kj = sel6(kv, j);
        acc += ki * kj;
#pragma unroll
        for (int t = 0; t < 6; ++t) ks[t] += (t == i ? kj : 0.f) + (t == j ? ki : 0.f);
      }
    }
#pragma unroll
    for (int t = 0; t < 6; ++t)
      if (t < pl.nt) Ksum[t * BB + e] = ks[t] * gscale;
  }
  float t = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) loss[blockIdx.x] = t / ((float)D * D);          // `loss` = this launch's partial array (one per block)
}

__global__ __launch_bounds__(256) void diff_finish_kernel(const float* __restrict__ dA, const float* __restrict__ invn, int B,
                                                          int D, int64_t stride, float* dx, const float* lparts, int nparts, float* loss) {
  const int k = blockIdx.x;
  if (k == 0 && threadIdx.x == 0 && loss) loss_parts_add(lparts, nparts, loss);
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < B; ++r) s += dA[((int64_t)k * B + r) * D + i] * invn[k * B + r];
    float cm = s / B;
    for (int r = 0; r < B; ++r) dx[k * stride + (int64_t)r * D + i] += dA[((int64_t)k * B + r) * D + i] * invn[k * B + r] - cm;
  }
}

// ------------------------------------------------------------------------------------------------ cmd
// One workgroup; thread per (tensor, column).  Moments k=1..5 per column, pair norms, then the analytic gradient
//   d/dx[r,i] = scale/(3B) * [U1[i] + sum_{k=2..5} U_k[i] * k * (s[r,i]^(k-1) - c_{k-1}[i])]
// with U_k = sum over pairs of (+/-) (m_a,k - m_b,k)/||m_a,k - m_b,k|| and c_1 = 0.
__global__ __launch_bounds__(256) void cmd_kernel(const float* __restrict__ x, int64_t stride, int B, int D, float scale,
                                                  float vscale, float* loss, float* dx, PairList pl, int nmom) {
  extern __shared__ float sm[];
  __shared__ float red[16];
  __shared__ float nrm[6][5];
  float* mom = sm;                 // [3][5][D]
  float* U = sm + 15 * D;          // [3][5][D]
  for (int it = threadIdx.x; it < pl.nt * D; it += blockDim.x) {
    int k = it / D, i = it % D;
    const float* X = x + k * stride;
    float s = 0.f;
    for (int r = 0; r < B; ++r) s += X[(int64_t)r * D + i];
    float m = s / B;
    float c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    for (int r = 0; r < B; ++r) {
      float d = X[(int64_t)r * D + i] - m;
      float d2 = d * d;
      c2 += d2; c3 += d2 * d; c4 += d2 * d2; c5 += d2 * d2 * d;
    }
    mom[(k * 5 + 0) * D + i] = m;
    mom[(k * 5 + 1) * D + i] = c2 / B;
    mom[(k * 5 + 2) * D + i] = c3 / B;
    mom[(k * 5 + 3) * D + i] = c4 / B;
    mom[(k * 5 + 4) * D + i] = c5 / B;
  }
  __syncthreads();
  float total = 0.f;
  for (int p = 0; p < pl.np; ++p)
    for (int k = 0; k < nmom; ++k) {
      float acc = 0.f;
      for (int i = threadIdx.x; i < D; i += blockDim.x) {
        float d = mom[(pl.a[p] * 5 + k) * D + i] - mom[(pl.b[p] * 5 + k) * D + i];
        acc += d * d;
      }
      float t = sqrtf(block_sum(acc, red));
      if (threadIdx.x == 0) nrm[p][k] = t;
      total += t;
    }
  __syncthreads();
  if (threadIdx.x == 0 && loss) atomicAdd(loss, total * vscale);
  if (!dx) return;
  for (int it = threadIdx.x; it < 15 * D; it += blockDim.x) U[it] = 0.f;
  __syncthreads();
  for (int it = threadIdx.x; it < nmom * D; it += blockDim.x) {
    int k = it / D, i = it % D;
    for (int p = 0; p < pl.np; ++p) {
      float d = mom[(pl.a[p] * 5 + k) * D + i] - mom[(pl.b[p] * 5 + k) * D + i];
      float u = d / nrm[p][k];              // 0/0 -> NaN exactly like sqrt'(0) in the reference
      U[(pl.a[p] * 5 + k) * D + i] += u;    // same thread owns column i for every p: no race
      U[(pl.b[p] * 5 + k) * D + i] -= u;
    }
  }
  __syncthreads();
  const float gs = scale * vscale / B;
  for (int it = threadIdx.x; it < pl.nt * D; it += blockDim.x) {
    int k = it / D, i = it % D;
    const float* X = x + k * stride;
    float m = mom[(k * 5) * D + i];
    float c2 = mom[(k * 5 + 1) * D + i], c3 = mom[(k * 5 + 2) * D + i], c4 = mom[(k * 5 + 3) * D + i];
    float u1 = U[(k * 5) * D + i], u2 = U[(k * 5 + 1) * D + i], u3 = U[(k * 5 + 2) * D + i];
    float u4 = U[(k * 5 + 3) * D + i], u5 = U[(k * 5 + 4) * D + i];
    for (int r = 0; r < B; ++r) {
      float d = X[(int64_t)r * D + i] - m;
      float d2 = d * d;
      float g = u1 + u2 * 2.f * d + u3 * 3.f * (d2 - c2) + u4 * 4.f * (d2 * d - c3) + u5 * 5.f * (d2 * d2 - c4);
      dx[k * stride + (int64_t)r * D + i] += gs * g;
    }
  }
}

// Parallel form of cmd_kernel for the training shapes (nt <= 3 tensors, D <= 128 columns, B <= 64 rows): 1024 threads =
// 128 columns x 8 row groups, every element is loaded once and stays in registers through the three passes (mean, central
// moments, gradient); cross-row reductions go through LDS.  cmd_kernel walks the rows serially per (tensor, column) with
// one dependent global load per row and pass, which cost 41 us of the step for 48 KB of input.
template <int CMD_RPT>                           // rows per thread: 4 (B <= 32) or 8 (B <= 64)
__device__ __forceinline__ void cmd_fast_body(const float* __restrict__ x, int64_t stride, int B, int D, float scale, float vscale,
                                              float* loss, float* dx, const PairList& pl, int nmom) {
  extern __shared__ float sm[];
  float* P1 = sm;                                // [8][3][128]
  float* P2 = sm + 8 * 3 * 128;                  // [8][3][4][128]
  float* N2 = P2 + 8 * 3 * 4 * 128;              // [2][16] per-wave partial squared norms
  const int tid = threadIdx.x, c = tid & 127, rg = tid >> 7;
  const int cc = min(c, D - 1);
  const bool c_ok = c < D;
  float v[3][CMD_RPT];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < CMD_RPT; ++j) {
      const int r = rg + 8 * j;
      const float val = x[(int64_t)(t < pl.nt ? t : 0) * stride + (int64_t)min(r, B - 1) * D + cc];
      v[t][j] = (c_ok && r < B && t < pl.nt) ? val : 0.f;
    }
  const float invB = 1.f / B;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CMD_RPT; ++j) s += v[t][j];
    P1[(rg * 3 + t) * 128 + c] = s;
  }
  __syncthreads();
  float mom[3][5];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += P1[(g * 3 + t) * 128 + c];
    mom[t][0] = s * invB;
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f;
#pragma unroll
    for (int j = 0; j < CMD_RPT; ++j) {
      const bool ok = c_ok && rg + 8 * j < B;
      const float d = ok ? v[t][j] - mom[t][0] : 0.f;
      const float d2 = d * d;
      q2 += d2; q3 += d2 * d; q4 += d2 * d2; q5 += d2 * d2 * d;
    }
    P2[((rg * 3 + t) * 4 + 0) * 128 + c] = q2; P2[((rg * 3 + t) * 4 + 1) * 128 + c] = q3;
    P2[((rg * 3 + t) * 4 + 2) * 128 + c] = q4; P2[((rg * 3 + t) * 4 + 3) * 128 + c] = q5;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += P2[((g * 3 + t) * 4 + k) * 128 + c];
      mom[t][k + 1] = s * invB;
    }
  // squared norms of the moment differences, per (pair, moment): the two waves of row group 0 hold one column per lane
  float dk[3][5];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) { a = (pl.a[p] == t) ? mom[t][k] : a; b = (pl.b[p] == t) ? mom[t][k] : b; }
      dk[p][k] = (p < pl.np && k < nmom && c_ok) ? a - b : 0.f;
    }
  if (rg == 0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float s = wave_sum(dk[p][k] * dk[p][k]);
        if ((tid & 63) == 0) N2[(tid >> 6) * 16 + p * 5 + k] = s;
      }
  }
  __syncthreads();
  float nrm[3][5], total = 0.f;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      nrm[p][k] = sqrtf(N2[p * 5 + k] + N2[16 + p * 5 + k]);
      if (p < pl.np && k < nmom) total += nrm[p][k];
    }
  if (tid == 0 && loss) atomicAdd(loss, total * vscale);
  if (!dx) return;
  float U[3][5];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 5; ++k) U[t][k] = 0.f;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (p < pl.np && k < nmom) {                       // uniform
        const float u = dk[p][k] / nrm[p][k];            // 0/0 -> NaN exactly like sqrt'(0) in the reference
#pragma unroll
        for (int t = 0; t < 3; ++t) U[t][k] += (pl.a[p] == t ? u : 0.f) - (pl.b[p] == t ? u : 0.f);
      }
    }
  const float gs = scale * vscale / B;
  float old[3][CMD_RPT];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < CMD_RPT; ++j)
      old[t][j] = dx[(int64_t)(t < pl.nt ? t : 0) * stride + (int64_t)min(rg + 8 * j, B - 1) * D + cc];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < CMD_RPT; ++j) {
      const int r = rg + 8 * j;
      if (!(c_ok && r < B && t < pl.nt)) continue;
      const float d = v[t][j] - mom[t][0];
      const float d2 = d * d;
      const float g = U[t][0] + U[t][1] * 2.f * d + U[t][2] * 3.f * (d2 - mom[t][1]) + U[t][3] * 4.f * (d2 * d - mom[t][2]) +
                      U[t][4] * 5.f * (d2 * d2 - mom[t][3]);
      dx[(int64_t)t * stride + (int64_t)r * D + c] = old[t][j] + gs * g;
    }
}

template <int CMD_RPT>
// tail_flag (optional): set to tail_value when the (single) workgroup's stores are out -- the launch is the last of a stream's chain
// and another stream's kernel waits for that word on the device (misa.hip: flag joins), instead of a one-thread launch behind it
__global__ __launch_bounds__(1024) void cmd_fast_kernel(const float* __restrict__ x, int64_t stride, int B, int D, float scale,
                                                        float vscale, float* loss, float* dx, PairList pl, int nmom,
                                                        unsigned* tail_flag, unsigned tail_value) {
  cmd_fast_body<CMD_RPT>(x, stride, B, D, scale, vscale, loss, dx, pl, nmom);
  if (tail_flag) {                                  // launch-uniform
    __syncthreads();                                // (every wave's stores have reached L2: vmcnt(0) in front of the barrier)
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(tail_flag, tail_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// cmd_fast_kernel for any row count (B = 256 per GPU is BASELINE's data-parallel shape): same 128 columns x 8 row groups and the
// same reductions, but the elements are re-read in each of the three passes (the inputs are <= 400 KB and sit in L2) instead of
// being kept in registers.  The serial cmd_kernel took 320 us of the B = 256 step.
__global__ __launch_bounds__(1024) void cmd_stream_kernel(const float* __restrict__ x, int64_t stride, int B, int D, float scale,
                                                          float vscale, float* loss, float* dx, PairList pl, int nmom) {
  extern __shared__ float sm[];
  float* P1 = sm;                                // [8][3][128]
  float* P2 = sm + 8 * 3 * 128;                  // [8][3][4][128]
  float* N2 = P2 + 8 * 3 * 4 * 128;              // [2][16] per-wave partial squared norms
  const int tid = threadIdx.x, c = tid & 127, rg = tid >> 7;
  const int cc = min(c, D - 1);
  const bool c_ok = c < D;
  const float invB = 1.f / B;
  // Every pass walks the rows in blocks of 64 (eight per thread) with the loads of ALL tensors of a block in flight together: the
  // inputs sit in L2, a pass is a chain of load latencies, and one tensor at a time at eight loads deep took 54 us at B = 256.  (The
  // additions of a tensor still run in row order: the same bits as before.)
  auto rows8 = [&](int r0, float (&v)[3][8]) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[t][j] = x[(int64_t)(t < pl.nt ? t : 0) * stride + (int64_t)min(r0 + 8 * j, B - 1) * D + cc];      // (clamped: no load under a branch)
  };
  {
    float s[3] = {0.f, 0.f, 0.f};
    for (int r0 = rg; r0 < B; r0 += 64) {
      float v[3][8];
      rows8(r0, v);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[t] += (r0 + 8 * j < B && t < pl.nt && c_ok) ? v[t][j] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) P1[(rg * 3 + t) * 128 + c] = s[t];
  }
  __syncthreads();
  float mom[3][5];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += P1[(g * 3 + t) * 128 + c];
    mom[t][0] = s * invB;
  }
  {
    float q[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) q[t][k] = 0.f;
    for (int r0 = rg; r0 < B; r0 += 64) {
      float v[3][8];
      rows8(r0, v);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (r0 + 8 * j < B && t < pl.nt && c_ok) ? v[t][j] - mom[t][0] : 0.f;
          const float d2 = d * d;
          q[t][0] += d2; q[t][1] += d2 * d; q[t][2] += d2 * d2; q[t][3] += d2 * d2 * d;
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) P2[((rg * 3 + t) * 4 + k) * 128 + c] = q[t][k];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += P2[((g * 3 + t) * 4 + k) * 128 + c];
      mom[t][k + 1] = s * invB;
    }
  float dk[3][5];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) { a = (pl.a[p] == t) ? mom[t][k] : a; b = (pl.b[p] == t) ? mom[t][k] : b; }
      dk[p][k] = (p < pl.np && k < nmom && c_ok) ? a - b : 0.f;
    }
  if (rg == 0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float s = wave_sum(dk[p][k] * dk[p][k]);
        if ((tid & 63) == 0) N2[(tid >> 6) * 16 + p * 5 + k] = s;
      }
  }
  __syncthreads();
  float nrm[3][5], total = 0.f;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      nrm[p][k] = sqrtf(N2[p * 5 + k] + N2[16 + p * 5 + k]);
      if (p < pl.np && k < nmom) total += nrm[p][k];
    }
  if (tid == 0 && loss && blockIdx.x == 0) atomicAdd(loss, total * vscale);      // (single adder: every workgroup holds the same total)
  if (!dx) return;
  float U[3][5];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 5; ++k) U[t][k] = 0.f;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (p < pl.np && k < nmom) {                       // uniform
        const float u = dk[p][k] / nrm[p][k];            // 0/0 -> NaN exactly like sqrt'(0) in the reference
#pragma unroll
        for (int t = 0; t < 3; ++t) U[t][k] += (pl.a[p] == t ? u : 0.f) - (pl.b[p] == t ? u : 0.f);
      }
    }
  const float gs = scale * vscale / B;
  if (gridDim.x > 1) {
    // One workgroup per tensor (large batches): every workgroup walked the moments of all tensors -- a single workgroup pulls its
    // inputs through ONE CU at ~45 GB/s, which made the whole kernel 57 us at B = 256 -- and now writes the gradient of its own.
    const int t = blockIdx.x;
    if (!(t < pl.nt && c_ok)) return;
    float ut[5], mt[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      ut[k] = t == 0 ? U[0][k] : (t == 1 ? U[1][k] : U[2][k]);
      mt[k] = t == 0 ? mom[0][k] : (t == 1 ? mom[1][k] : mom[2][k]);
    }
    for (int r0 = rg; r0 < B; r0 += 64) {
      float v[8], old[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t o = (int64_t)t * stride + (int64_t)min(r0 + 8 * j, B - 1) * D + cc;
        v[j] = x[o]; old[j] = dx[o];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (r0 + 8 * j >= B) continue;
        const float d = v[j] - mt[0];
        const float d2 = d * d;
        const float g = ut[0] + ut[1] * 2.f * d + ut[2] * 3.f * (d2 - mt[1]) + ut[3] * 4.f * (d2 * d - mt[2]) + ut[4] * 5.f * (d2 * d2 - mt[3]);
        dx[(int64_t)t * stride + (int64_t)(r0 + 8 * j) * D + c] = old[j] + gs * g;
      }
    }
    return;
  }
  for (int r0 = rg; r0 < B; r0 += 64) {
    float v[3][8], old[3][8];
    rows8(r0, v);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        old[t][j] = dx[(int64_t)(t < pl.nt ? t : 0) * stride + (int64_t)min(r0 + 8 * j, B - 1) * D + cc];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!(r0 + 8 * j < B && t < pl.nt && c_ok)) continue;
        const float d = v[t][j] - mom[t][0];
        const float d2 = d * d;
        const float g = U[t][0] + U[t][1] * 2.f * d + U[t][2] * 3.f * (d2 - mom[t][1]) + U[t][3] * 4.f * (d2 * d - mom[t][2]) +
                        U[t][4] * 5.f * (d2 * d2 - mom[t][3]);
        dx[(int64_t)t * stride + (int64_t)(r0 + 8 * j) * D + c] = old[t][j] + gs * g;
      }
  }
}

// ------------------------------------------------------------------------------------------------ recon (MSE)
__global__ __launch_bounds__(256) void recon_kernel(const float* __restrict__ rec, const float* __restrict__ orig, int64_t n,
                                                    float inv_n, float scale, float* loss, float* drec, float* dorig) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float d = rec[e] - orig[e];
    acc += d * d;
    float g = 2.f * d * inv_n * scale;
    if (drec) drec[e] += g;
    if (dorig) dorig[e] -= g;
  }
  float t = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) loss[blockIdx.x] = t * inv_n;                   // `loss` = this launch's partial array (one per block)
}

// ------------------------------------------------------------------------------------------------ domain (CE)
__global__ __launch_bounds__(256) void domain_kernel(const float* __restrict__ dom, int B, float scale, float* loss, float* ddom) {
  __shared__ float red[16];
  float acc = 0.f;
  const int rows = 3 * B;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    int label = r / B;
    float z0 = dom[r * 3], z1 = dom[r * 3 + 1], z2 = dom[r * 3 + 2];
    float m = fmaxf(z0, fmaxf(z1, z2));
    float e0 = expf(z0 - m), e1 = expf(z1 - m), e2 = expf(z2 - m);
    float se = e0 + e1 + e2;
    float lse = m + logf(se);
    float zl = label == 0 ? z0 : (label == 1 ? z1 : z2);
    acc += lse - zl;
    if (ddom) {
      float g = scale / rows;
      ddom[r * 3 + 0] += g * (e0 / se - (label == 0 ? 1.f : 0.f));
      ddom[r * 3 + 1] += g * (e1 / se - (label == 1 ? 1.f : 0.f));
      ddom[r * 3 + 2] += g * (e2 / se - (label == 2 ? 1.f : 0.f));
    }
  }
  float t = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, t / rows);
}

// ------------------------------------------------------------------------------------------------ evaluation counts
// Per-class tp / fp / fn over (pred > 0, truth > 0) and the Jaccard-style accuracy sum of the reference's get_accuracy
// (utils/eval.py:14-31), accumulated into a device state so that an evaluation pass over many batches needs one read-back:
//   state[0..C) tp, [C..2C) fp, [2C..3C) fn, [3C] sum_i |y_i & p_i| / max(|y_i | p_i|, 1), [3C+1] samples      (doubles)
constexpr int EVAL_MAXC = 16;
__global__ __launch_bounds__(256) void eval_counts_kernel(const float* __restrict__ pred, const float* __restrict__ truth, int N, int C,
                                                          double* state) {
  double tp[EVAL_MAXC], fp[EVAL_MAXC], fn[EVAL_MAXC];
#pragma unroll
  for (int c = 0; c < EVAL_MAXC; ++c) { tp[c] = 0.0; fp[c] = 0.0; fn[c] = 0.0; }
  double jac = 0.0, cnt = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    int inter = 0, uni = 0;
#pragma unroll
    for (int c = 0; c < EVAL_MAXC; ++c) {
      if (c < C) {
        const bool y = truth[(int64_t)i * C + c] > 0.f, p = pred[(int64_t)i * C + c] > 0.f;
        tp[c] += (y && p) ? 1.0 : 0.0; fp[c] += (!y && p) ? 1.0 : 0.0; fn[c] += (y && !p) ? 1.0 : 0.0;
        inter += (y && p) ? 1 : 0; uni += (y || p) ? 1 : 0;
      }
    }
    jac += (double)inter / (double)(uni > 0 ? uni : 1);
    cnt += 1.0;
  }
  auto wsum = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  const bool lead = (threadIdx.x & 63) == 0;
#pragma unroll
  for (int c = 0; c < EVAL_MAXC; ++c) {
    if (c < C) {
      const double a = wsum(tp[c]), b = wsum(fp[c]), d = wsum(fn[c]);
      if (lead) { if (a != 0.0) atomicAdd(&state[c], a); if (b != 0.0) atomicAdd(&state[C + c], b); if (d != 0.0) atomicAdd(&state[2 * C + c], d); }
    }
  }
  jac = wsum(jac); cnt = wsum(cnt);
  if (lead && cnt != 0.0) { atomicAdd(&state[3 * C], jac); atomicAdd(&state[3 * C + 1], cnt); }
}

}  // namespace

extern "C" int mmda_heads_fwd(const float* logits, int B, int ncls, float threshold, float* tcp, float* scores, float* labels,
                              float drop_p, uint64_t seed, int site, void* stream) {
  if (!logits || !tcp || !scores || !labels || B <= 0 || ncls <= 0) return MMDA_EINVAL;
  int n = B * (6 + ncls);
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, B, ncls, threshold, tcp,
                     scores, labels, drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_heads_fwd");
  return MMDA_OK;
}

extern "C" int mmda_heads_bwd(const float* tcp, const float* scores, const float* dtcp, const float* dscores, int B, int ncls,
                              float* dlogits, float drop_p, uint64_t seed, int site, void* stream) {
  if (!tcp || !scores || !dlogits || B <= 0 || ncls <= 0) return MMDA_EINVAL;
  int n = B * (6 + ncls);
  hipLaunchKernelGGL(heads_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, tcp, scores, dtcp, dscores, B,
                     ncls, dlogits, drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_heads_bwd");
  return MMDA_OK;
}

extern "C" int mmda_loss_cls(const float* scores, const float* emo, int B, int ncls, float scale, float* loss, float* dscores,
                             void* stream) {
  if (!scores || !emo || B <= 0 || ncls <= 0) return MMDA_EINVAL;
  hipLaunchKernelGGL(cls_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, emo, B, ncls, scale, loss, dscores);
  MMDA_CHECK_LAUNCH("mmda_loss_cls");
  return MMDA_OK;
}

extern "C" int mmda_loss_conf(const float* scores, const float* tcp, const float* emo, int B, int ncls, float scale, float* loss,
                              float* dscores, float* dtcp, void* stream) {
  if (!scores || !tcp || !emo || B <= 0 || ncls != 6) return MMDA_EINVAL;   // tcp has 6 columns (models.py:139)
  float* parts = loss ? mmda_scratch_get((hipStream_t)stream, sizeof(float) * ncls) : nullptr;
  if (loss && !parts) return MMDA_ELAUNCH;
  hipLaunchKernelGGL(conf_kernel, dim3(ncls), dim3(256), 0, (hipStream_t)stream, scores, tcp, emo, B, ncls, scale, parts, dscores, dtcp);
  if (loss) hipLaunchKernelGGL(loss_parts_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, parts, ncls, loss);
  MMDA_CHECK_LAUNCH("mmda_loss_conf");
  return MMDA_OK;
}

extern "C" int64_t mmda_loss_diff_work_floats(int B, int D) {
  return (int64_t)12 * B * D + 6 * B + 6 * D + (int64_t)12 * B * B;
}

namespace {
bool fill_pairs(PairList& pl, int nt, int np, const int* pairs, int max_t) {
  if (nt < 2 || nt > max_t || np < 1 || np > 6 || !pairs) return false;
  pl.nt = nt; pl.np = np;
  for (int p = 0; p < 6; ++p) { pl.a[p] = 0; pl.b[p] = 0; }
  for (int p = 0; p < np; ++p) {
    pl.a[p] = pairs[2 * p]; pl.b[p] = pairs[2 * p + 1];
    if (pl.a[p] < 0 || pl.a[p] >= nt || pl.b[p] < 0 || pl.b[p] >= nt) return false;
  }
  return true;
}
}  // namespace

extern "C" int mmda_loss_diff_pairs(const float* x, int64_t stride, int nt, int np, const int* pairs_host, int B, int D, float scale,
                                    float* loss, float* dx, float* work, void* stream) {
  PairList pl;
  if (!x || !work || B <= 0 || D <= 0 || !fill_pairs(pl, nt, np, pairs_host, 6)) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* Ahat = work;                               // 6*B*D
  float* dA = Ahat + (int64_t)6 * B * D;            // 6*B*D
  float* invn = dA + (int64_t)6 * B * D;            // 6*B
  float* mean = invn + 6 * B;                       // 6*D
  float* K = mean + 6 * D;                          // 6*B*B
  float* Ksum = K + (int64_t)6 * B * B;             // 6*B*B
  const bool fast = D <= 128 && B <= 256;            // rows per thread 4 / 8 / 16 / 32 (B <= 32 / 64 / 128 / 256)
  if (fast) {
    if (B <= 32) hipLaunchKernelGGL(diff_prep_fast_kernel<4>, dim3(nt), dim3(1024), 0, s, x, stride, B, D, Ahat, invn, mean);
    else if (B <= 64) hipLaunchKernelGGL(diff_prep_fast_kernel<8>, dim3(nt), dim3(1024), 0, s, x, stride, B, D, Ahat, invn, mean);
    else if (B <= 128) hipLaunchKernelGGL(diff_prep_fast_kernel<16>, dim3(nt), dim3(1024), 0, s, x, stride, B, D, Ahat, invn, mean);
    else hipLaunchKernelGGL(diff_prep_fast_kernel<32>, dim3(nt), dim3(1024), 0, s, x, stride, B, D, Ahat, invn, mean);
  } else {
    hipLaunchKernelGGL(diff_prep_kernel, dim3(nt), dim3(256), 0, s, x, stride, B, D, Ahat, invn, mean);
  }
  MMDA_CHECK_LAUNCH("mmda_loss_diff/prep");
  const int64_t BB = (int64_t)B * B;
  int rc;
  if (B <= 256) {
    // few rows: the nt Gram matrices K_k = Ahat_k Ahat_k^T as one row-skinny launch
    mmda_skinny_args sk[6];
    for (int k = 0; k < nt; ++k) {
      sk[k] = mmda_skinny_args{};
      sk[k].M = B; sk[k].N = B; sk[k].K = D; sk[k].transB = 1; sk[k].A = Ahat + (int64_t)k * B * D; sk[k].lda = D;
      sk[k].B = Ahat + (int64_t)k * B * D; sk[k].ldb = D; sk[k].C = K + k * BB; sk[k].ldc = B;
    }
    rc = mmda_gemm_skinny(sk, nt, stream);
  } else {
    mmda_gemm_args g = {};
    g.mode = MMDA_F32; g.transA = 0; g.transB = 1; g.M = B; g.N = B; g.K = D; g.batch = nt;
    g.A = Ahat; g.lda = D; g.strideA = (int64_t)B * D;
    g.B = Ahat; g.ldb = D; g.strideB = (int64_t)B * D;
    g.C = K; g.ldc = B; g.strideC = (int64_t)B * B;
    rc = mmda_gemm(&g, stream);
  }
  if (rc) return rc;
  int blocks = (int)((BB + 255) / 256); if (blocks > 256) blocks = 256;
  // the loss value: one partial per block (per-stream scratch), added in block order -- by the finish launch below when it follows with
  // nothing but a row-skinny GEMM in between (which takes no scratch), else by a launch of its own right here
  float* lparts = loss ? mmda_scratch_get(s, sizeof(float) * 256) : nullptr;
  if (loss && !lparts) return MMDA_ELAUNCH;
  hipLaunchKernelGGL(diff_combine_kernel, dim3(blocks), dim3(256), 0, s, K, B, D, scale, lparts, Ksum, pl);
  MMDA_CHECK_LAUNCH("mmda_loss_diff/combine");
  const bool sum_later = loss && dx && B <= 256;
  if (loss && !sum_later) {
    hipLaunchKernelGGL(loss_parts_finish_kernel, dim3(1), dim3(64), 0, s, lparts, blocks, loss);
    MMDA_CHECK_LAUNCH("mmda_loss_diff/loss");
  }
  float* const loss_later = sum_later ? loss : nullptr;
  if (!dx) return MMDA_OK;
  if (B <= 256) {
    mmda_skinny_args sk[6];
    for (int k = 0; k < nt; ++k) {
      sk[k] = mmda_skinny_args{};
      sk[k].M = B; sk[k].N = D; sk[k].K = B; sk[k].transB = 0; sk[k].A = Ksum + k * BB; sk[k].lda = B;
      sk[k].B = Ahat + (int64_t)k * B * D; sk[k].ldb = D; sk[k].C = dA + (int64_t)k * B * D; sk[k].ldc = D;
    }
    rc = mmda_gemm_skinny(sk, nt, stream);
  } else {
    mmda_gemm_args h = {};
    h.mode = MMDA_F32; h.transA = 0; h.transB = 0; h.M = B; h.N = D; h.K = B; h.batch = nt;
    h.A = Ksum; h.lda = B; h.strideA = BB;
    h.B = Ahat; h.ldb = D; h.strideB = (int64_t)B * D;
    h.C = dA; h.ldc = D; h.strideC = (int64_t)B * D;
    rc = mmda_gemm(&h, stream);
  }
  if (rc) return rc;
  if (fast) {
    if (B <= 32) hipLaunchKernelGGL(diff_finish_fast_kernel<4>, dim3(nt), dim3(1024), 0, s, dA, invn, B, D, stride, dx, lparts, blocks, loss_later);
    else if (B <= 64) hipLaunchKernelGGL(diff_finish_fast_kernel<8>, dim3(nt), dim3(1024), 0, s, dA, invn, B, D, stride, dx, lparts, blocks, loss_later);
    else if (B <= 128) hipLaunchKernelGGL(diff_finish_fast_kernel<16>, dim3(nt), dim3(1024), 0, s, dA, invn, B, D, stride, dx, lparts, blocks, loss_later);
    else hipLaunchKernelGGL(diff_finish_fast_kernel<32>, dim3(nt), dim3(1024), 0, s, dA, invn, B, D, stride, dx, lparts, blocks, loss_later);
  } else {
    hipLaunchKernelGGL(diff_finish_kernel, dim3(nt), dim3(256), 0, s, dA, invn, B, D, stride, dx, lparts, blocks, loss_later);
  }
  MMDA_CHECK_LAUNCH("mmda_loss_diff/finish");
  return MMDA_OK;
}

extern "C" int mmda_loss_diff(const float* x, int64_t stride, int B, int D, float scale, float* loss, float* dx, float* work,
                              void* stream) {
  // the six pairs of solver.py:432-439 over [private_t, private_v, private_a, shared_t, shared_v, shared_a]
  static const int pairs[12] = {0, 3, 1, 4, 2, 5, 2, 0, 2, 1, 0, 1};
  return mmda_loss_diff_pairs(x, stride, 6, 6, pairs, B, D, scale, loss, dx, work, stream);
}

// internal (misa.hip): arm the NEXT mmda_loss_cmd* call of this thread to set *flag = value at the end of its launch -- honoured by the
// single-workgroup form only (mmda_loss_cmd_sets_flag says whether a call of that shape will)
static thread_local unsigned* g_cmd_tail_flag = nullptr;
static thread_local unsigned g_cmd_tail_value = 0u;
bool mmda_loss_cmd_sets_flag(int B, int D) { return D <= 128 && B <= 64; }
void mmda_loss_cmd_arm_flag(unsigned* flag, unsigned value) { g_cmd_tail_flag = flag; g_cmd_tail_value = value; }

extern "C" int mmda_loss_cmd_pairs(const float* x, int64_t stride, int nt, int np, const int* pairs_host, int n_moments, int B, int D,
                                   float scale, float value_scale, float* loss, float* dx, void* stream) {
  unsigned* tflag = g_cmd_tail_flag;
  const unsigned tvalue = g_cmd_tail_value;
  g_cmd_tail_flag = nullptr;
  PairList pl;
  if (!x || B <= 0 || D <= 0 || n_moments < 1 || n_moments > 5 || !fill_pairs(pl, nt, np, pairs_host, 3)) return MMDA_EINVAL;
  if (nt <= 3 && D <= 128 && B <= 64) {
    const size_t lds_fast = sizeof(float) * (8 * 3 * 128 + 8 * 3 * 4 * 128 + 32);
    if (B <= 32)
      hipLaunchKernelGGL(cmd_fast_kernel<4>, dim3(1), dim3(1024), lds_fast, (hipStream_t)stream, x, stride, B, D, scale, value_scale,
                         loss, dx, pl, n_moments, tflag, tvalue);
    else
      hipLaunchKernelGGL(cmd_fast_kernel<8>, dim3(1), dim3(1024), lds_fast, (hipStream_t)stream, x, stride, B, D, scale, value_scale,
                         loss, dx, pl, n_moments, tflag, tvalue);
    MMDA_CHECK_LAUNCH("mmda_loss_cmd");
    return MMDA_OK;
  }
  if (nt <= 3 && D <= 128) {
    const size_t lds_fast = sizeof(float) * (8 * 3 * 128 + 8 * 3 * 4 * 128 + 32);
    hipLaunchKernelGGL(cmd_stream_kernel, dim3(dx ? nt : 1), dim3(1024), lds_fast, (hipStream_t)stream, x, stride, B, D, scale, value_scale, loss,
                       dx, pl, n_moments);
    MMDA_CHECK_LAUNCH("mmda_loss_cmd");
    return MMDA_OK;
  }
  size_t lds = sizeof(float) * 30 * D;
  if (lds > 60 * 1024) return MMDA_EINVAL;
  hipLaunchKernelGGL(cmd_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, x, stride, B, D, scale, value_scale, loss, dx, pl,
                     n_moments);
  MMDA_CHECK_LAUNCH("mmda_loss_cmd");
  return MMDA_OK;
}

extern "C" int mmda_loss_cmd(const float* x, int64_t stride, int B, int D, float scale, float* loss, float* dx, void* stream) {
  static const int pairs[6] = {0, 1, 0, 2, 2, 1};        // (t,v), (t,a), (a,v)   solver.py:415-417, then /3
  return mmda_loss_cmd_pairs(x, stride, 3, 3, pairs, 5, B, D, scale, 1.0f / 3.0f, loss, dx, stream);
}

extern "C" int mmda_loss_recon(const float* recon, const float* orig, int64_t stride, int B, int D, float scale, float* loss,
                               float* drecon, float* dorig, void* stream) {
  if (!recon || !orig || B <= 0 || D <= 0) return MMDA_EINVAL;
  // three (B,D) tensors at +k*stride; mean over each, averaged over the three (solver.py:445-448).  All three have
  // the same element count, so sum(d^2)/(3*B*D) over the lot is the same number: one launch when they are contiguous.
  const int64_t n = (int64_t)B * D;
  const int nl = (stride == n) ? 1 : 3;
  for (int k = 0; k < nl; ++k) {
    int64_t cnt = (nl == 1) ? 3 * n : n;
    int blocks = (int)((cnt + 255) / 256); if (blocks > 64) blocks = 64;
    float* parts = loss ? mmda_scratch_get((hipStream_t)stream, sizeof(float) * 64) : nullptr;
    if (loss && !parts) return MMDA_ELAUNCH;
    hipLaunchKernelGGL(recon_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, recon + k * stride, orig + k * stride, cnt,
                       1.0f / (3.0f * n), scale, parts, drecon ? drecon + k * stride : nullptr, dorig ? dorig + k * stride : nullptr);
    if (loss) hipLaunchKernelGGL(loss_parts_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, parts, blocks, loss);
  }
  MMDA_CHECK_LAUNCH("mmda_loss_recon");
  return MMDA_OK;
}

extern "C" int mmda_loss_misc(const float* scores, const float* tcp, const float* emo, int B, int ncls, float* d_scores, float* d_tcp,
                              int with_conf, int conf_grads, float conf_scale, const float* recon, const float* orig, int64_t n_recon,
                              float recon_scale, float* d_recon, float* d_orig, float* L, float diff_w, float sim_w, float recon_w,
                              float conf_w, int use_conf, void* stream) {
  if (!scores || !emo || !recon || !orig || !L || B <= 0 || ncls <= 0 || n_recon <= 0) return MMDA_EINVAL;
  if (with_conf && (!tcp || ncls != 6)) return MMDA_EINVAL;
  MiscLossArgs a = {};
  a.scores = scores; a.tcp = tcp; a.emo = emo; a.B = B; a.ncls = ncls; a.d_scores = d_scores; a.d_tcp = d_tcp;
  a.conf_grads = conf_grads && d_scores && d_tcp; a.conf_scale = conf_scale; a.with_conf = with_conf;
  a.recon = recon; a.orig = orig; a.n_recon = n_recon; a.recon_inv_n = 1.0f / (float)n_recon; a.recon_scale = recon_scale;
  a.d_recon = d_recon; a.d_orig = d_orig; a.L = L; a.dw = diff_w; a.sw = sim_w; a.rw = recon_w; a.cw = conf_w; a.use_conf = use_conf;
  // (eight elements per thread at least: the last workgroup adds the partials one after the other, and 3 x 32 x 128 elements do not
  //  need 64 of them)
  int rb = (int)((n_recon + 2047) / 2048); if (rb > 64) rb = 64; if (rb < 1) rb = 1;
  a.recon_blocks = rb;
  const int blocks = 1 + (with_conf ? ncls : 0) + rb;
  a.parts = mmda_scratch_get((hipStream_t)stream, sizeof(float) * blocks);
  if (!a.parts) return MMDA_ELAUNCH;
  hipLaunchKernelGGL(misc_losses_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  MMDA_CHECK_LAUNCH("mmda_loss_misc");
  return MMDA_OK;
}

extern "C" int mmda_eval_accumulate(const float* pred, const float* truth, int N, int C, double* state, void* stream) {
  if (!pred || !truth || !state || N < 0 || C <= 0 || C > EVAL_MAXC || ((uintptr_t)state & 7)) return MMDA_EINVAL;
  if (N == 0) return MMDA_OK;
  int blocks = (N + 255) / 256; if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(eval_counts_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, truth, N, C, state);
  MMDA_CHECK_LAUNCH("mmda_eval_accumulate");
  return MMDA_OK;
}

extern "C" int mmda_loss_domain(const float* dom, int B, float scale, float* loss, float* ddom, void* stream) {
  if (!dom || B <= 0) return MMDA_EINVAL;
  hipLaunchKernelGGL(domain_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dom, B, scale, loss, ddom);
  MMDA_CHECK_LAUNCH("mmda_loss_domain");
  return MMDA_OK;
}
