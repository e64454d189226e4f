// Row-wise kernels: LayerNorm forward/backward (wave-per-row shuffle reductions), embedding gather / scatter-add,
// small elementwise helpers.  All HBM-bound; loads are lane-consecutive (coalesced 256 B per wave instruction).
#include "common.h"
#include <stdlib.h>
#include "rowlocal.h"
#include "splitk.h"

int mmda_embed_scatter_sorted(float* dW, const int64_t* ids, int n, int D, const float* rows, const int* lengths, int B, void* stream);   // dist.hip
int mmda_embed_scatter_add_masked(float* dW, const int64_t* ids, int rows, int dim, const float* dX, const int* lengths, int B, void* stream);
bool mmda_embed_scatter_sorts(int rows);

namespace {

constexpr int LN_MAXQ = 16;   // up to 16*64 = 1024 columns cached in registers per lane

constexpr int LN_MAXP = 4;      // problems per launch (the three modalities' LayerNorms go out together)
struct LnMulti { mmda_ln_args a[LN_MAXP]; int start[LN_MAXP + 1]; int n; };
// part[p]: (nblk[p], 2, n) partial gamma / beta gradients of problem p, one row pair per block (ln_bwd_kernel) or row chunk
// (ln_param_grads_kernel); ln_pg_finish_kernel adds them in block order -- no float atomics, identical bits on every run
struct LnBwdMulti { mmda_ln_bwd_args a[LN_MAXP]; int start[LN_MAXP + 1]; int nblk[LN_MAXP]; float* part[LN_MAXP]; int n; };

// NQ = values per lane kept in registers (64 NQ >= n): the loops below are fully unrolled over it, so a launch is instantiated
// for the smallest NQ that covers its widest problem (n = 128 needs 2; the generic 16 costs 8x the instructions)
template <int NQ>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnMulti L) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < LN_MAXP; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_ln_args& a = L.a[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = ((int)blockIdx.x - L.start[pi]) * 4 + wave;
  if (row >= a.rows) return;
  ln_fwd_row<NQ>(a, row, lane);      // (rowlocal.h)
}

template <int NQ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdMulti L) {
  __shared__ float red[2][4][NQ * 64];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < LN_MAXP; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_ln_bwd_args& a = L.a[pi];
  const int blk = (int)blockIdx.x - L.start[pi], nblk = L.nblk[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.n;
  const bool want_pg = a.dgamma != nullptr || a.dbeta != nullptr;
  float dg[NQ], db[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { dg[q] = 0.f; db[q] = 0.f; }
  for (int row = blk * 4 + wave; row < a.rows; row += nblk * 4) ln_bwd_row<NQ>(a, row, lane, dg, db);      // (rowlocal.h)
  if (!want_pg) return;            // block-uniform: parameter gradients come from mmda_layernorm_param_grads instead
  // reduce the per-wave column partials across the block's 4 waves (fixed order): the block's partial row pair
  const int nq = (n + 63) / 64;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q < nq) { red[0][wave][q * 64 + lane] = dg[q]; red[1][wave][q * 64 + lane] = db[q]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float g = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
    float b = red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i];
    L.part[pi][((int64_t)blk * 2 + 0) * n + i] = g;
    L.part[pi][((int64_t)blk * 2 + 1) * n + i] = b;
  }
}

// dgamma[i] += the blocks' partials, in a FIXED order: sixteen threads per output take every sixteenth partial each (their loads are
// independent), and the sixteen sub-sums are added in lane order
__global__ __launch_bounds__(256) void ln_pg_finish_kernel(LnBwdMulti L) {
  __shared__ float sub[256];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < LN_MAXP; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_ln_bwd_args& a = L.a[pi];
  const int n = a.n;
  const int e = ((int)blockIdx.x - L.start[pi]) * 16 + (threadIdx.x >> 4);       // (gamma | beta, column)
  const int j = threadIdx.x & 15;
  const bool ok = e < 2 * n;
  const int which = ok ? e / n : 0, i = ok ? e % n : 0;
  const float* p = L.part[pi] + (int64_t)which * n + i;
  const int64_t stride = 2 * (int64_t)n;
  const int nb = L.nblk[pi];
  float acc = 0.f;
  if (ok) {
    int q = j;
    for (; q + 7 * 16 < nb; q += 8 * 16) {                 // eight loads in flight (one at a time is a cache latency per partial)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(q + 16 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; q < nb; q += 16) acc += p[q * stride];
  }
  sub[threadIdx.x] = acc;
  __syncthreads();
  float* dst = which ? a.dbeta : a.dgamma;
  if (ok && j == 0 && dst) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += sub[(threadIdx.x & ~15) + u];
    dst[i] += t;
  }
}

// ------------------------------------------------------------------------------------------------ 16-byte forms
// The inter-layer LayerNorms (rows = T * B, n = 2H: 600 / 148 wide; plain: no activation, residual or permutation) with 16-byte
// accesses: a wave per row, lane l holds the float4 groups l, l + 64, ... of the row (n a multiple of 4, n <= 1024).  The scalar forms
// above move 4 bytes per lane and instruction and ran at 2 TB/s at B = 256 (ln_fwd 40 us, ln_bwd 68 us for 84 / 126 MB).
// Rows whose width is even but not a multiple of four (the visual encoder: 70) take the same path with 8-byte groups.
constexpr int LNV_MAX = 4;         // groups per lane: n <= 1024 (16-byte groups)
template <int VW> struct VecOf;
template <> struct VecOf<4> { typedef float T __attribute__((ext_vector_type(4))); };
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <int VW> __device__ __forceinline__ float vsum(typename VecOf<VW>::T v) {
  if (VW == 4) return (v[0] + v[1]) + (v[2] + v[3]);
  return v[0] + v[1];
}
template <int VW> __device__ __forceinline__ typename VecOf<VW>::T vzero() {
  typename VecOf<VW>::T z;
#pragma unroll
  for (int e = 0; e < VW; ++e) z[e] = 0.f;
  return z;
}

template <int NV, int VW>
__device__ __forceinline__ void ln_fwd_vec_row(const mmda_ln_args& a, int row, int lane) {
  typedef typename VecOf<VW>::T V;
  const int n = a.n, ng = n / VW;
  const V* x = reinterpret_cast<const V*>(a.x + (int64_t)row * n);
  V v[NV];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int g = lane + 64 * q;
    v[q] = g < ng ? x[g] : vzero<VW>();
    s += vsum<VW>(v[q]);
  }
  const float mean = wave_sum(s) / n;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    if (lane + 64 * q < ng) {
      const V d = v[q] - mean;
      ss += vsum<VW>(d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / n + a.eps);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
  const V* gm = reinterpret_cast<const V*>(a.gamma);
  const V* bt = reinterpret_cast<const V*>(a.beta);
  V* y = a.y ? reinterpret_cast<V*>(a.y + (int64_t)row * n) : nullptr;
  unsigned short* yb = a.y_bf16 ? reinterpret_cast<unsigned short*>(a.y_bf16) + (int64_t)row * a.ld_bf16 : nullptr;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int g = lane + 64 * q;
    V o = vzero<VW>();
    if (g < ng) { o = (v[q] - mean) * rstd * gm[g] + bt[g]; if (y) y[g] = o; }
    if (yb && VW * g < a.ld_bf16) {                                     // (ld_bf16 is a multiple of 8: whole groups; zero in the padding)
      if (VW == 4) {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        const u2v pk = {(unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16), (unsigned)f2bf(o[2 % VW]) | ((unsigned)f2bf(o[3 % VW]) << 16)};
        *reinterpret_cast<u2v*>(yb + 4 * g) = pk;
      } else {
        *reinterpret_cast<unsigned*>(yb + 2 * g) = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
      }
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(LnMulti L) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < LN_MAXP; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_ln_args& a = L.a[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = ((int)blockIdx.x - L.start[pi]) * 4 + wave;
  if (row >= a.rows) return;
  if (a.n & 3) ln_fwd_vec_row<NV, 2>(a, row, lane);        // problem-uniform
  else ln_fwd_vec_row<NV, 4>(a, row, lane);
}

// backward, d_x of every row plus (parts != null) the block's partial gamma / beta gradients: block `blk` of a problem leaves
// parts[(blk * 2 + {0,1}) * n + i]; mmda_ln_parts_finish adds the blocks in order (no atomics).
template <int NV, int VW>
__device__ __forceinline__ void ln_bwd_vec_body(const mmda_ln_bwd_args& a, int blk, int nblk, float* part, float* red_raw) {
  typedef typename VecOf<VW>::T V;
  V (*red)[4][NV * 64] = reinterpret_cast<V (*)[4][NV * 64]>(red_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.n, ng = n / VW;
  const V* gm = reinterpret_cast<const V*>(a.gamma);
  V gam[NV], dg[NV], db[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int g = lane + 64 * q;
    gam[q] = g < ng ? gm[g] : vzero<VW>();
    dg[q] = vzero<VW>(); db[q] = vzero<VW>();
  }
  for (int row = blk * 4 + wave; row < a.rows; row += nblk * 4) {
    const float mean = a.mean[row], rstd = a.rstd[row];
    const V* x = reinterpret_cast<const V*>(a.x + (int64_t)row * n);
    const V* dy = reinterpret_cast<const V*>(a.dy + (int64_t)row * n);
    V xh[NV], gdy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int g = lane + 64 * q;
      xh[q] = vzero<VW>(); gdy[q] = vzero<VW>();
      if (g < ng) {
        const V d = dy[g];
        xh[q] = (x[g] - mean) * rstd;
        gdy[q] = d * gam[q];
        dg[q] += d * xh[q];
        db[q] += d;
        s1 += vsum<VW>(gdy[q]);
        s2 += vsum<VW>(gdy[q] * xh[q]);
      }
    }
    s1 = wave_sum(s1) / n;
    s2 = wave_sum(s2) / n;
    V* dx = reinterpret_cast<V*>(a.d_x + (int64_t)row * n);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int g = lane + 64 * q;
      if (g < ng) {
        V d = rstd * (gdy[q] - s1 - xh[q] * s2);
        if (a.accumulate_dx) d += dx[g];
        dx[g] = d;
      }
    }
  }
  if (!part) return;                                       // problem-uniform
#pragma unroll
  for (int q = 0; q < NV; ++q) { red[0][wave][q * 64 + lane] = dg[q]; red[1][wave][q * 64 + lane] = db[q]; }
  __syncthreads();
  for (int g = threadIdx.x; g < ng; g += 256) {
    const V gs = (red[0][0][g] + red[0][1][g]) + (red[0][2][g] + red[0][3][g]);
    const V bs = (red[1][0][g] + red[1][1][g]) + (red[1][2][g] + red[1][3][g]);
    *reinterpret_cast<V*>(part + ((int64_t)blk * 2 + 0) * n + VW * g) = gs;
    *reinterpret_cast<V*>(part + ((int64_t)blk * 2 + 1) * n + VW * g) = bs;
  }
}
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(LnBwdMulti L) {
  __shared__ __attribute__((aligned(16))) float red[2 * 4 * NV * 64 * 4];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < LN_MAXP; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_ln_bwd_args& a = L.a[pi];
  const int blk = (int)blockIdx.x - L.start[pi];
  if (a.n & 3) ln_bwd_vec_body<NV, 2>(a, blk, L.nblk[pi], L.part[pi], red);      // problem-uniform
  else ln_bwd_vec_body<NV, 4>(a, blk, L.nblk[pi], L.part[pi], red);
}

// dgamma / dbeta alone, parallel over column strips x row chunks: lane = column (coalesced 256-B rows), no wave reductions,
// one partial per column and row chunk (added in chunk order by ln_pg_finish_kernel).  For the three big inter-layer LayerNorms (rows = T*B) the
// fused form above spends most of its time in ~200-way contended atomics; this pass runs off the critical path instead.
__global__ __launch_bounds__(256) void ln_param_grads_kernel(LnBwdMulti L) {
  __shared__ float red[2][4][64];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < LN_MAXP; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_ln_bwd_args& a = L.a[pi];
  const int blk = (int)blockIdx.x - L.start[pi];
  const int n = a.n, strips = (n + 63) / 64;
  const int strip = blk % strips, chunk = blk / strips, nchunk = L.nblk[pi];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = strip * 64 + lane;
  const int ic = min(i, n - 1);
  const int per = (a.rows + nchunk - 1) / nchunk;
  const int r0 = chunk * per, r1 = min(a.rows, r0 + per);
  float dg = 0.f, db = 0.f;
  for (int row = r0 + wave; row < r1; row += 4) {
    const float mean = a.mean[row], rstd = a.rstd[row];
    const int64_t drow = perm_row(row, a.permute_S, a.permute_B);
    const int64_t idx = (int64_t)row * n + ic;
    float x = act_fwd_p(a.act, a.x[idx], a.actp, (uint64_t)idx);
    if (a.res) x += a.res[idx] * drop_mul(a.drop_p, a.drop_seed, a.drop_site, (uint64_t)idx);
    const float dy = a.dy[drow * n + ic];
    dg += dy * (x - mean) * rstd;
    db += dy;
  }
  red[0][wave][lane] = dg; red[1][wave][lane] = db;
  __syncthreads();
  if (wave == 0 && i < n) {
    const float g = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    const float b = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    L.part[pi][((int64_t)chunk * 2 + 0) * n + i] = g;
    L.part[pi][((int64_t)chunk * 2 + 1) * n + i] = b;
  }
}

__global__ void embed_gather_kernel(const float* __restrict__ W, const int64_t* __restrict__ ids, int rows, int dim, float* out) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = W + ids[row] * (int64_t)dim;
  float* dst = out + (int64_t)row * dim;
  for (int i = threadIdx.x & 63; i < dim; i += 64) dst[i] = src[i];
}

// dW[ids[p]] += dX[p] for every position p, deterministically: one workgroup per position; the workgroup of the FIRST position that
// holds an id owns that id -- it marks every position with the same id in an LDS bit mask (one pass over the id list), compacts the
// marks into an ordered list, and its four waves add the rows of four contiguous quarters of that list in position order (eight row
// loads in flight per lane and dimension group), the quarters' sums are added in order and the total goes to the table row (single
// writer: no atomics, identical bits on every run; float atomics gave the rows of repeated ids in arrival order).  Lists of up to
// ES_MAX positions (longer ones take the sort-based path of dist.hip).  `lengths` (optional, with B): position p = t * B + b is padding
// when t >= lengths[b] -- its row is exactly zero (no gradient flows through padding) and it is skipped, as owner and as contributor:
// a ragged batch holds the pad id hundreds of times.
constexpr int ES_MAX = 4096;
__global__ __launch_bounds__(256) void embed_scatter_kernel(float* dW, const int64_t* __restrict__ ids, int rows, int dim, const float* __restrict__ dX,
                                                            const int* __restrict__ lengths, int B) {
  __shared__ unsigned mask[ES_MAX / 32];
  __shared__ unsigned short list[ES_MAX];
  __shared__ int base[ES_MAX / 32 + 1];
  __shared__ int earlier;
  __shared__ float part[3][1024];
  const int p = blockIdx.x;
  const int64_t id = ids[p];
  auto padded = [&](int q) { return lengths != nullptr && (q / B) >= lengths[q % B]; };
  if (id < 0 || padded(p)) return;                       // block-uniform
  const int nwords = (rows + 31) / 32;
  if (threadIdx.x == 0) earlier = 0;
  for (int i = threadIdx.x; i < nwords; i += 256) mask[i] = 0u;
  __syncthreads();
  // one pass over the ids, eight loads in flight per thread
#pragma unroll 8
  for (int q = threadIdx.x; q < rows; q += 256)
    if (ids[q] == id && !padded(q)) atomicOr(&mask[q >> 5], 1u << (q & 31));
  __syncthreads();
  for (int i = threadIdx.x; i < (p + 31) / 32; i += 256) {      // an earlier position with this id owns it
    unsigned m = mask[i];
    if (32 * i + 32 > p) m &= (1u << (p - 32 * i)) - 1u;
    if (m) earlier = 1;                                  // benign race: every writer stores 1
  }
  __syncthreads();
  if (earlier) return;                                   // block-uniform
  // ordered list of the positions >= p that hold the id
  if (threadIdx.x == 0) {
    int c = 0;
    for (int i = p >> 5; i < nwords; ++i) {
      base[i] = c;
      unsigned m = mask[i];
      if (i == (p >> 5)) m &= ~((1u << (p & 31)) - 1u);
      c += __builtin_popcount(m);
    }
    base[nwords] = c;
  }
  __syncthreads();
  const int total = base[nwords];
  for (int i = (p >> 5) + threadIdx.x; i < nwords; i += 256) {
    unsigned m = mask[i];
    if (i == (p >> 5)) m &= ~((1u << (p & 31)) - 1u);
    int o = base[i];
    while (m) { list[o++] = (unsigned short)(32 * i + __builtin_ctz(m)); m &= m - 1; }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (total + 3) / 4;
  const int q0 = min(total, wave * per), q1 = min(total, q0 + per);
  for (int d0 = 0; d0 < dim; d0 += 256) {                // dimension groups of 256: lane handles d0 + lane + 64 j, j < 4
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = q0; q < q1; q += 8) {
      float v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t r = list[min(q + u, q1 - 1)];      // (clamped: no load under a branch; the surplus is masked below)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d = d0 + lane + 64 * j;
          v[u][j] = dX[r * dim + min(d, dim - 1)];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (q + u < q1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += v[u][j];
        }
    }
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) part[wave - 1][lane + 64 * j] = acc[j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = d0 + lane + 64 * j;
        if (d < dim) dW[id * (int64_t)dim + d] += ((acc[j] + part[0][lane + 64 * j]) + part[1][lane + 64 * j]) + part[2][lane + 64 * j];
      }
    }
    __syncthreads();
  }
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

__global__ void act_drop_fwd_kernel(const float* z, float* h, int64_t n, int act, mmda_act_params ap, float p, uint64_t seed, int site) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    h[i] = act_fwd_p(act, z[i], ap, (uint64_t)i) * drop_mul(p, seed, site, (uint64_t)i);
}
__global__ __launch_bounds__(256) void act_drop_bwd_kernel(const float* dh, const float* z, float* dz, int64_t n, int act, mmda_act_params ap,
                                                           float p, uint64_t seed, int site) {
  __shared__ float red[16];
  float dslope = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = dh[i] * drop_mul(p, seed, site, (uint64_t)i);
    const float zi = z[i];
    dz[i] = g * act_bwd_p(act, zi, ap, (uint64_t)i);
    if (act == MMDA_ACT_PRELU && zi <= 0.f) dslope += g * zi;
  }
  if (act == MMDA_ACT_PRELU && ap.dslope) {              // launch-uniform
    const float t = block_sum(dslope, red);
    if (threadIdx.x == 0) atomicAdd(ap.dslope, t);
  }
}

int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

namespace {
int ln_check(const mmda_ln_args* a) {
  if (!a->x || (!a->y && !a->y_bf16) || !a->gamma || !a->beta || a->rows < 0 || a->n <= 0 || a->n > LN_MAXQ * 64) return MMDA_EINVAL;      // (y may be NULL when only the bf16 copy is wanted)
  if (a->permute_S > 0 && (a->permute_B <= 0 || a->permute_S * a->permute_B != a->rows)) return MMDA_EINVAL;
  return MMDA_OK;
}
int ln_bwd_check(const mmda_ln_bwd_args* a) {
  if (!a->dy || !a->x || !a->gamma || !a->mean || !a->rstd || a->rows < 0 || a->n <= 0 || a->n > LN_MAXQ * 64) return MMDA_EINVAL;
  if (a->permute_S > 0 && (a->permute_B <= 0 || a->permute_S * a->permute_B != a->rows)) return MMDA_EINVAL;
  return MMDA_OK;
}
}  // namespace

namespace {
// partial buffers of the problems that want parameter gradients (per-stream scratch, api.hip) ...
int ln_pg_alloc(LnBwdMulti& L, hipStream_t s) {
  int64_t floats = 0;
  for (int k = 0; k < L.n; ++k)
    if (L.a[k].dgamma || L.a[k].dbeta) floats += (int64_t)L.nblk[k] * 2 * L.a[k].n;
  for (int k = 0; k < LN_MAXP; ++k) L.part[k] = nullptr;
  if (floats == 0) return MMDA_OK;
  float* base = mmda_scratch_get(s, sizeof(float) * (size_t)floats);
  if (!base) return MMDA_ELAUNCH;
  for (int k = 0; k < L.n; ++k)
    if (L.a[k].dgamma || L.a[k].dbeta) { L.part[k] = base; base += (int64_t)L.nblk[k] * 2 * L.a[k].n; }
  return MMDA_OK;
}
// ... and the launch that adds them up, behind the launch that wrote them (same stream)
int ln_pg_finish(const LnBwdMulti& Lw, hipStream_t s) {
  LnBwdMulti F = Lw;
  int blocks = 0, n = 0;
  for (int k = 0; k < Lw.n; ++k) {
    if (!Lw.part[k]) continue;
    F.a[n] = Lw.a[k]; F.nblk[n] = Lw.nblk[k]; F.part[n] = Lw.part[k]; F.start[n] = blocks; blocks += ceil_div(2 * Lw.a[k].n, 16); n++;
  }
  if (n == 0) return MMDA_OK;
  F.n = n;
  for (int k = n; k <= LN_MAXP; ++k) F.start[k] = blocks;
  for (int k = n; k < LN_MAXP; ++k) { F.a[k] = F.a[0]; F.nblk[k] = 1; F.part[k] = F.part[0]; }
  hipLaunchKernelGGL(ln_pg_finish_kernel, dim3(blocks), dim3(256), 0, s, F);
  MMDA_CHECK_LAUNCH("mmda_layernorm_param_grads(finish)");
  return MMDA_OK;
}
}  // namespace

extern "C" int mmda_layernorm_fwd_multi(const mmda_ln_args* a, int n, void* stream) {
  if (!a || n < 0) return MMDA_EINVAL;
  for (int i = 0; i < n; ++i)
    if (ln_check(a + i)) return MMDA_EINVAL;
  for (int base = 0; base < n; base += LN_MAXP) {
    LnMulti L;
    L.n = 0;
    int blocks = 0;
    for (int i = base; i < n && i < base + LN_MAXP; ++i) {
      if (a[i].rows == 0) continue;
      L.a[L.n] = a[i]; L.start[L.n] = blocks; blocks += ceil_div(a[i].rows, 4); L.n++;
    }
    for (int k = L.n; k <= LN_MAXP; ++k) L.start[k] = blocks;
    for (int k = L.n; k < LN_MAXP; ++k) L.a[k] = L.a[0];
    if (blocks == 0) continue;
    int nq = 1;
    for (int k = 0; k < L.n; ++k) nq = max(nq, ceil_div(L.a[k].n, 64));
    bool vec = nq > 2;                                   // (the 128-wide LayerNorms of the fusion block stay on the scalar form)
    for (int k = 0; k < L.n; ++k) {
      const mmda_ln_args& q = L.a[k];
      const int vw = (q.n & 3) ? 2 : 4;                 // bytes per group 8 / 16: every row start must be that aligned
      vec = vec && (q.n & 1) == 0 && ceil_div(q.n, vw * 64) <= LNV_MAX && !q.res && q.act == MMDA_ACT_NONE && q.permute_S <= 0 &&
            ((((uintptr_t)q.x | (uintptr_t)q.y | (uintptr_t)q.gamma | (uintptr_t)q.beta) & (4 * vw - 1)) == 0) &&      // (a NULL y is aligned)
            (!q.y_bf16 || ((q.ld_bf16 & 7) == 0 && ((uintptr_t)q.y_bf16 & 7) == 0 && ceil_div(q.ld_bf16, vw * 64) <= LNV_MAX));
    }
    static const int ln_vec_on = getenv("MMDA_LN_VEC") ? atoi(getenv("MMDA_LN_VEC")) : 1;
    if (vec && ln_vec_on) {
      int nv = 1;                                        // groups per lane: the widest problem in ITS group size (bf16 copy included)
      for (int k = 0; k < L.n; ++k) {
        const int vw = (L.a[k].n & 3) ? 2 : 4;
        nv = max(nv, ceil_div(max(L.a[k].n, L.a[k].y_bf16 ? L.a[k].ld_bf16 : 0), vw * 64));
      }
      if (nv <= 1) hipLaunchKernelGGL(ln_fwd_vec_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
      else if (nv <= 2) hipLaunchKernelGGL(ln_fwd_vec_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
      else if (nv <= 3) hipLaunchKernelGGL(ln_fwd_vec_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
      else hipLaunchKernelGGL(ln_fwd_vec_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
      MMDA_CHECK_LAUNCH("mmda_layernorm_fwd(vec)");
      continue;
    }
    if (nq <= 2) hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    else if (nq <= 4) hipLaunchKernelGGL(ln_fwd_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    else if (nq <= 10) hipLaunchKernelGGL(ln_fwd_kernel<10>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    else hipLaunchKernelGGL(ln_fwd_kernel<LN_MAXQ>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    MMDA_CHECK_LAUNCH("mmda_layernorm_fwd");
  }
  return MMDA_OK;
}
extern "C" int mmda_layernorm_fwd(const mmda_ln_args* a, void* stream) { return mmda_layernorm_fwd_multi(a, a ? 1 : -1, stream); }

extern "C" int mmda_layernorm_bwd_multi(const mmda_ln_bwd_args* a, int n, void* stream) {
  if (!a || n < 0) return MMDA_EINVAL;
  for (int i = 0; i < n; ++i)
    if (ln_bwd_check(a + i)) return MMDA_EINVAL;
  for (int base = 0; base < n; base += LN_MAXP) {
    LnBwdMulti L;
    L.n = 0;
    int blocks = 0;
    for (int i = base; i < n && i < base + LN_MAXP; ++i) {
      if (a[i].rows == 0) continue;
      const bool pg = a[i].dgamma || a[i].dbeta;
      // with parameter gradients: 2 rows per wave keeps the per-column atomics at rows/8 adders; without: one row per wave
      int nb = ceil_div(a[i].rows, pg ? 8 : 4);
      if (nb > 1024) nb = 1024;
      L.a[L.n] = a[i]; L.start[L.n] = blocks; L.nblk[L.n] = nb; blocks += nb; L.n++;
    }
    for (int k = L.n; k <= LN_MAXP; ++k) L.start[k] = blocks;
    for (int k = L.n; k < LN_MAXP; ++k) { L.a[k] = L.a[0]; L.nblk[k] = 1; }
    if (blocks == 0) continue;
    if (ln_pg_alloc(L, (hipStream_t)stream)) return MMDA_ELAUNCH;
    int nq = 1;
    for (int k = 0; k < L.n; ++k) nq = max(nq, ceil_div(L.a[k].n, 64));
    if (nq <= 2) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    else if (nq <= 4) hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    else if (nq <= 10) hipLaunchKernelGGL(ln_bwd_kernel<10>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    else hipLaunchKernelGGL(ln_bwd_kernel<LN_MAXQ>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    MMDA_CHECK_LAUNCH("mmda_layernorm_bwd");
    if (ln_pg_finish(L, (hipStream_t)stream)) return MMDA_ELAUNCH;
  }
  return MMDA_OK;
}
extern "C" int mmda_layernorm_bwd(const mmda_ln_bwd_args* a, void* stream) { return mmda_layernorm_bwd_multi(a, a ? 1 : -1, stream); }

// Internal (misa.hip): the backward of up to LN_MAXP plain LayerNorms with the gamma / beta gradients left as per-block partials in a
// caller-owned buffer (`parts`, mmda_ln_parts_floats floats) -- the d_x launch on the critical stream computes them on the way (it
// reads dy and x anyway), and mmda_ln_parts_finish adds them up on whatever stream the caller likes (behind an event).  Replaces
// the separate column-strip pass over dy and x (mmda_layernorm_param_grads: 100 us at B = 256) when the 16-byte form applies.
namespace {
bool ln_bwd_vec_applies(const mmda_ln_bwd_args& q) {
  const int vw = (q.n & 3) ? 2 : 4;
  return (q.n & 1) == 0 && ceil_div(q.n, vw * 64) <= LNV_MAX && !q.res && !q.d_res && q.act == MMDA_ACT_NONE && q.permute_S <= 0 && q.d_x &&
         ((((uintptr_t)q.x | (uintptr_t)q.dy | (uintptr_t)q.gamma | (uintptr_t)q.d_x) & (4 * vw - 1)) == 0);
}
int ln_parts_blocks(const mmda_ln_bwd_args& q) { int nb = ceil_div(q.rows, 16); return nb > 512 ? 512 : (nb < 1 ? 1 : nb); }
}  // namespace
bool mmda_ln_bwd_parts_applies(const mmda_ln_bwd_args* a, int n) {
  static const int ln_vec_on = getenv("MMDA_LN_VEC") ? atoi(getenv("MMDA_LN_VEC")) : 1;
  if (!ln_vec_on || !a || n <= 0 || n > LN_MAXP) return false;
  for (int i = 0; i < n; ++i)
    if (ln_bwd_check(a + i) || !ln_bwd_vec_applies(a[i])) return false;
  return true;
}
int64_t mmda_ln_parts_floats(const mmda_ln_bwd_args* a, int n) {
  int64_t f = 0;
  for (int i = 0; i < n; ++i) f += (int64_t)ln_parts_blocks(a[i]) * 2 * a[i].n;
  return f;
}
static void ln_parts_layout(const mmda_ln_bwd_args* a, int n, float* parts, LnBwdMulti& L, int& blocks) {
  L.n = 0; blocks = 0;
  for (int i = 0; i < n; ++i) {
    const int nb = ln_parts_blocks(a[i]);
    L.a[L.n] = a[i]; L.start[L.n] = blocks; L.nblk[L.n] = nb; L.part[L.n] = parts; parts += (int64_t)nb * 2 * a[i].n;
    blocks += nb; L.n++;
  }
  for (int k = L.n; k <= LN_MAXP; ++k) L.start[k] = blocks;
  for (int k = L.n; k < LN_MAXP; ++k) { L.a[k] = L.a[0]; L.nblk[k] = 1; L.part[k] = L.part[0]; }
}
int mmda_ln_bwd_parts(const mmda_ln_bwd_args* a, int n, float* parts, void* stream) {
  if (!mmda_ln_bwd_parts_applies(a, n) || !parts) return MMDA_EINVAL;
  LnBwdMulti L;
  int blocks = 0;
  ln_parts_layout(a, n, parts, L, blocks);
  int ng = 1;
  for (int k = 0; k < L.n; ++k) ng = max(ng, ceil_div(L.a[k].n, ((L.a[k].n & 3) ? 2 : 4) * 64));
  if (ng <= 1) hipLaunchKernelGGL(ln_bwd_vec_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  else if (ng <= 2) hipLaunchKernelGGL(ln_bwd_vec_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  else if (ng <= 3) hipLaunchKernelGGL(ln_bwd_vec_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  else hipLaunchKernelGGL(ln_bwd_vec_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  MMDA_CHECK_LAUNCH("mmda_ln_bwd_parts");
  return MMDA_OK;
}
int mmda_ln_parts_finish(const mmda_ln_bwd_args* a, int n, float* parts, void* stream) {
  if (!a || n <= 0 || n > LN_MAXP || !parts) return MMDA_EINVAL;
  LnBwdMulti L;
  int blocks = 0;
  ln_parts_layout(a, n, parts, L, blocks);
  return ln_pg_finish(L, (hipStream_t)stream);
}

extern "C" int mmda_layernorm_param_grads(const mmda_ln_bwd_args* a, int n, void* stream) {
  if (!a || n < 0) return MMDA_EINVAL;
  for (int i = 0; i < n; ++i)
    if (ln_bwd_check(a + i) || (!a[i].dgamma && !a[i].dbeta)) return MMDA_EINVAL;
  for (int base = 0; base < n; base += LN_MAXP) {
    LnBwdMulti L;
    L.n = 0;
    int blocks = 0;
    for (int i = base; i < n && i < base + LN_MAXP; ++i) {
      if (a[i].rows == 0) continue;
      const int strips = ceil_div(a[i].n, 64);
      int chunks = ceil_div(a[i].rows, 32);          // 8 rows per wave
      if (chunks * strips > 1024) chunks = (1024 + strips - 1) / strips;
      L.a[L.n] = a[i]; L.start[L.n] = blocks; L.nblk[L.n] = chunks; blocks += chunks * strips; L.n++;
    }
    for (int k = L.n; k <= LN_MAXP; ++k) L.start[k] = blocks;
    for (int k = L.n; k < LN_MAXP; ++k) { L.a[k] = L.a[0]; L.nblk[k] = 1; }
    if (blocks == 0) continue;
    if (ln_pg_alloc(L, (hipStream_t)stream)) return MMDA_ELAUNCH;
    hipLaunchKernelGGL(ln_param_grads_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    MMDA_CHECK_LAUNCH("mmda_layernorm_param_grads");
    if (ln_pg_finish(L, (hipStream_t)stream)) return MMDA_ELAUNCH;
  }
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
  if (!dW || !ids || !dX || rows < 0 || dim <= 0 || dim > 1024) return MMDA_EINVAL;
  if (rows == 0) return MMDA_OK;
  // long lists: sort-based list-order sums (dist.hip) -- the scan below costs rows^2 / 256 id compares (B=256, T=50: 158 us against
  // 9 us at B=32).  MMDA_SCATTER_SORT_MIN moves the limit.
  return mmda_embed_scatter_add_masked(dW, ids, rows, dim, dX, nullptr, 0, stream);
}

// (internal) lists of this length take the sort-based form
bool mmda_embed_scatter_sorts(int rows) {
  static const int sort_min_env = getenv("MMDA_SCATTER_SORT_MIN") ? atoi(getenv("MMDA_SCATTER_SORT_MIN")) : 3072;
  const int sort_min = sort_min_env > ES_MAX ? ES_MAX + 1 : sort_min_env;
  return rows >= sort_min;
}
// (internal, misa.hip) the same with the batch's lengths: positions p = t * B + b with t >= lengths[b] are padding and are skipped
int mmda_embed_scatter_add_masked(float* dW, const int64_t* ids, int rows, int dim, const float* dX, const int* lengths, int B, void* stream) {
  if (!dW || !ids || !dX || rows < 0 || dim <= 0 || dim > 1024 || (lengths && B <= 0)) return MMDA_EINVAL;
  if (rows == 0) return MMDA_OK;
  if (mmda_embed_scatter_sorts(rows)) return mmda_embed_scatter_sorted(dW, ids, rows, dim, dX, lengths, B, stream);
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dW, ids, rows, dim, dX, lengths, B);
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

extern "C" int mmda_act_dropout_fwd_p(const float* z, float* h, int64_t n, int act, const mmda_act_params* ap, float drop_p, uint64_t seed,
                                      int site, void* stream) {
  if (!z || !h || n < 0) return MMDA_EINVAL;
  if ((act == MMDA_ACT_PRELU && (!ap || !ap->slope)) || (act == MMDA_ACT_RRELU && !ap)) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  const mmda_act_params p = ap ? *ap : mmda_act_params{};
  hipLaunchKernelGGL(act_drop_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, z, h, n, act, p, drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_act_dropout_fwd");
  return MMDA_OK;
}
extern "C" int mmda_act_dropout_fwd(const float* z, float* h, int64_t n, int act, float drop_p, uint64_t seed, int site, void* stream) {
  return mmda_act_dropout_fwd_p(z, h, n, act, nullptr, drop_p, seed, site, stream);
}

extern "C" int mmda_act_dropout_bwd_p(const float* dh, const float* z, float* dz, int64_t n, int act, const mmda_act_params* ap, float drop_p,
                                      uint64_t seed, int site, void* stream) {
  if (!dh || !z || !dz || n < 0) return MMDA_EINVAL;
  if ((act == MMDA_ACT_PRELU && (!ap || !ap->slope)) || (act == MMDA_ACT_RRELU && !ap)) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  const mmda_act_params p = ap ? *ap : mmda_act_params{};
  hipLaunchKernelGGL(act_drop_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dh, z, dz, n, act, p, drop_p, seed, site);
  MMDA_CHECK_LAUNCH("mmda_act_dropout_bwd");
  return MMDA_OK;
}
extern "C" int mmda_act_dropout_bwd(const float* dh, const float* z, float* dz, int64_t n, int act, float drop_p, uint64_t seed,
                                    int site, void* stream) {
  return mmda_act_dropout_bwd_p(dh, z, dz, n, act, nullptr, drop_p, seed, site, stream);
}
