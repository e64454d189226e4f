// Generic LDS-tiled MFMA GEMM for gfx950: fp32 data in HBM, f32 MFMA (exact) or bf16 MFMA (fp32 accumulate).
// One kernel serves every nn.Linear forward (NT), its input gradient (NN) and weight gradient (TN) on the path;
// operands whose inner dimension is 35/74/140/296/300 are handled by bounds-checked staging (zero fill).
//
// Tile 64x64x32, 256 threads = 4 waves (2x2), each wave a 32x32 sub-tile = 2x2 MFMA 16x16 accumulators.
#include "common.h"
#include "splitk.h"
#include <stdlib.h>
#include <vector>

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDS_F32_LD = 34;   // floats per LDS row: (2*row + k) % 32 distinct for the 16x16x4 operand reads
constexpr int LDS_BF16_LD = 40;  // shorts per LDS row (80 B, 16-B aligned rows for ds_read_b128)

template <int MODE, int PF>
__device__ __forceinline__ void gemm_body(const mmda_gemm_args& g, int splitk, unsigned char* smem, int bx, int by, int bzz,
                                          float* slab, int ldn) {
  float* As_f = reinterpret_cast<float*>(smem);
  float* Bs_f = As_f + BM * LDS_F32_LD;
  unsigned short* As_h = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bs_h = As_h + BM * LDS_BF16_LD;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int row0 = by * BM, col0 = bx * BN;
  const int64_t bz = bzz / splitk;
  const int sp = bzz % splitk;
  const float* A = g.A + bz * g.strideA;
  const float* A2 = g.A2 ? g.A2 + bz * g.strideA : nullptr;
  const float* Bm = g.B + bz * g.strideB;
  float* C = g.C + bz * g.strideC;
  const int M = g.M, N = g.N, K = g.K;

  // per-thread staging coordinates (fixed across k-tiles)
  int a_r[8], a_k[8], b_r[8], b_k[8];
  int64_t a_row_off_c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int e = tid + 256 * i;
    if (g.transA) { a_r[i] = e & 63; a_k[i] = e >> 6; } else { a_r[i] = e >> 5; a_k[i] = e & 31; }
    if (g.transB) { b_r[i] = e >> 5; b_k[i] = e & 31; } else { b_r[i] = e & 63; b_k[i] = e >> 6; }
    int m = row0 + a_r[i];
    int64_t src = min(m, M - 1);
    if (g.gather) src = g.gather[min(m, M - 1)];
    a_row_off_c[i] = src;                               // (clamped: rows past M load a valid row and are zeroed at the LDS store)
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Register staging.  Every load is UNCONDITIONAL from a clamped (always valid) address and the zero fill of the out-of-range
  // elements happens when the value is stored to LDS: a load under a lane-dependent branch costs a full `s_waitcnt vmcnt(0)` each --
  // sixteen serialised memory latencies per k-tile (the fusion block's weight-gradient launch took 232 us at B = 256 that way).
  float rap[PF][8], rbp[PF][8];
  unsigned okp[PF];                                     // bit i: A element i in range; bit 8 + i: B element i; bit 16 + i: B element i is the ones column
  auto load_tile = [&](float (&ra)[8], float (&rb)[8], unsigned& ok, int k0) {
    ok = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = row0 + a_r[i], k = k0 + a_k[i];
      const bool va = m < M && k < K;
      const int kc = max(0, min(k, K - 1));
      const int64_t off = g.transA ? (int64_t)kc * g.lda + min(m, M - 1) : a_row_off_c[i] * g.lda + kc;
      float v = A[off];
      if (A2) v += A2[off];                              // (kernel-uniform)
      ra[i] = v;
      ok |= va ? (1u << i) : 0u;
      const int n = col0 + b_r[i];
      const int kb = k0 + b_k[i];
      const bool vb = n < N && kb < K;
      const int nc = min(n, N - 1), kbc = max(0, min(kb, K - 1));
      rb[i] = g.transB ? Bm[(int64_t)nc * g.ldb + kbc] : Bm[(int64_t)kbc * g.ldb + nc];
      ok |= vb ? (1u << (8 + i)) : 0u;
      ok |= (g.bias_grad && n == N && kb < K) ? (1u << (16 + i)) : 0u;      // virtual all-ones column: its output is the bias gradient
    }
  };
  auto store_tile = [&](const float (&ra)[8], const float (&rb)[8], unsigned ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float av = (ok >> i) & 1u ? ra[i] : 0.f;
      const float bv = (ok >> (16 + i)) & 1u ? 1.0f : ((ok >> (8 + i)) & 1u ? rb[i] : 0.f);
      if (MODE == MMDA_BF16) {
        As_h[a_r[i] * LDS_BF16_LD + a_k[i]] = f2bf(av);
        Bs_h[b_r[i] * LDS_BF16_LD + b_k[i]] = f2bf(bv);
      } else {
        As_f[a_r[i] * LDS_F32_LD + a_k[i]] = av;
        Bs_f[b_r[i] * LDS_F32_LD + b_k[i]] = bv;
      }
    }
  };

  // split-K: this block reduces k-tiles [kt0, kt1); its raw partial tile goes into a slab of its own and the reduce launch behind
  // this one sums the slabs in slice order (splitk.h): no float atomics, identical bits on every run
  const int nk_all = (K + BK - 1) / BK;
  const int per = (nk_all + splitk - 1) / splitk;
  const int kt0 = sp * per;
  const int nk = min(nk_all, kt0 + per);
  if (splitk > 1 && kt0 >= nk) return;
  // PF == 1: one k-tile of register prefetch under the MFMAs (long k-loops, many workgroups in flight).
  // PF == 4: tiny problems (<= 4 k-tiles per block, a handful of blocks): issue ALL global loads up front so the block pays
  //          one first-touch memory latency instead of one per k-tile.
  if (PF > 1) {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (kt0 + p < nk) load_tile(rap[p], rbp[p], okp[p], (kt0 + p) * BK);
  } else {
    load_tile(rap[0], rbp[0], okp[0], kt0 * BK);
  }
#pragma unroll 1
  for (int kt = kt0; kt < nk; ++kt) {
    __syncthreads();            // previous tile's fragment reads are done
    if (PF > 1) {
      const int p = kt - kt0;
      if (p == 0) store_tile(rap[0], rbp[0], okp[0]);
      else if (p == 1) store_tile(rap[PF > 1 ? 1 : 0], rbp[PF > 1 ? 1 : 0], okp[PF > 1 ? 1 : 0]);
      else if (p == 2) store_tile(rap[PF > 2 ? 2 : 0], rbp[PF > 2 ? 2 : 0], okp[PF > 2 ? 2 : 0]);
      else store_tile(rap[PF > 3 ? 3 : 0], rbp[PF > 3 ? 3 : 0], okp[PF > 3 ? 3 : 0]);
    } else {
      store_tile(rap[0], rbp[0], okp[0]);
    }
    __syncthreads();
    if (PF == 1 && kt + 1 < nk) load_tile(rap[0], rbp[0], okp[0], (kt + 1) * BK);   // global loads fly under the MFMAs below
    const int fr = lane & 15, fq = lane >> 4;
    if (MODE == MMDA_BF16) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(&As_h[(wm * 32 + i * 16 + fr) * LDS_BF16_LD + fq * 8]);
        b[i] = *reinterpret_cast<const bf16x8*>(&Bs_h[(wn * 32 + i * 16 + fr) * LDS_BF16_LD + fq * 8]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = As_f[(wm * 32 + i * 16 + fr) * LDS_F32_LD + kk * 4 + fq];
          b[i] = Bs_f[(wn * 32 + i * 16 + fr) * LDS_F32_LD + kk * 4 + fq];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // epilogue: C/D fragment map col = lane&15, row = (lane>>4)*4 + reg
  if (splitk > 1) {                // host guarantees: no act/dropout/gate; slab = [batch][slice][M][ldn]
    float* S = slab + ((int64_t)bz * splitk + sp) * M * ldn;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = col0 + wn * 32 + j * 16 + (lane & 15);
        if (n >= ldn) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = row0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
          if (m < M) S[(int64_t)m * ldn + n] = acc[i][j][r];
        }
      }
    return;
  }
  const float* bias = g.bias ? g.bias + bz * g.strideBias : nullptr;
  const float* bias2 = g.bias2 ? g.bias2 + bz * g.strideBias : nullptr;
  const float alpha = g.alpha == 0.f ? 1.f : g.alpha;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = col0 + wn * 32 + j * 16 + (lane & 15);
      if (g.bias_grad && n == N) {                           // column sums of A (= dY) into the bias gradient(s): one writer per entry
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int m = row0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
          if (m < M) {
            g.bias_grad[bz * g.strideBias + m] += acc[i][j][r];
            if (g.bias_grad2) g.bias_grad2[bz * g.strideBias + m] += acc[i][j][r];
          }
        }
        continue;
      }
      if (n >= N) continue;
      float bsum = 0.f;
      if (bias) bsum += bias[n];
      if (bias2) bsum += bias2[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = row0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        int64_t ci = (int64_t)m * g.ldc + n;
        float v = alpha * acc[i][j][r] + bsum;
        if (g.accumulate) v += C[ci];
        v = act_fwd(g.act, v);
        if (g.drop_p > 0.f) v *= drop_mul(g.drop_p, g.drop_seed, g.drop_site, ((uint64_t)bz * M + m) * N + n);
        if (g.gate) v *= (g.gate[bz * g.strideC + (int64_t)m * g.ldgate + n] > 0.f) ? g.gate_scale : 0.f;
        C[ci] = v;
      }
    }
}


template <int MODE, int PF>
__global__ __launch_bounds__(256) void gemm_kernel(mmda_gemm_args g, int splitk, float* slab, int ldn) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * LDS_F32_LD * 4];
  gemm_body<MODE, PF>(g, splitk, smem, blockIdx.x, blockIdx.y, blockIdx.z, slab, ldn);
}

// Grouped launch: up to GROUP_MAX independent GEMMs (different shapes, layouts, modes) in ONE grid, so that the many small
// LSTM gradient GEMMs of the three modalities fill the chip together instead of running one after another at a few dozen
// workgroups each.  blockIdx.x -> (problem, tile) through the prefix table.
constexpr int GROUP_MAX = 16;
struct GroupLaunch {
  mmda_gemm_args p[GROUP_MAX];
  int start[GROUP_MAX + 1];
  int tx[GROUP_MAX], ty[GROUP_MAX], splitk[GROUP_MAX];
  float* slab[GROUP_MAX];
  int ldn[GROUP_MAX];
  int n;
};
__global__ __launch_bounds__(256) void gemm_grouped_kernel(GroupLaunch G) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * LDS_F32_LD * 4];
  int i = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < G.n && (int)blockIdx.x >= G.start[k]) i = k;
  const int local = blockIdx.x - G.start[i];
  const int bx = local % G.tx[i], by = (local / G.tx[i]) % G.ty[i], bzz = local / (G.tx[i] * G.ty[i]);
  const mmda_gemm_args g = G.p[i];      // one copy into SGPRs; indexing the kernarg array inside the k-loop would re-load fields
  const int sk = G.splitk[i];
  if (g.mode == MMDA_BF16) gemm_body<MMDA_BF16, 1>(g, sk, smem, bx, by, bzz, G.slab[i], G.ldn[i]);
  else gemm_body<MMDA_F32, 1>(g, sk, smem, bx, by, bzz, G.slab[i], G.ldn[i]);
}

// ------------------------------------------------------------------------------------------------ 128x128 bf16 tile
// The LSTM-sized GEMMs (M = T*B rows, N/K in {300, 600, 1200, 2400}) with 16-byte fp32 staging loads, a 128x128x32 tile
// and a 64x64 sub-tile per wave: 16 MFMA 16x16x32 per wave and k-tile against 8 ds_read_b128 (the 64x64 kernel above
// issues 4 MFMAs per 4 reads and 16 scalar loads per thread).  Requires 16-byte aligned rows (leading dims % 4 == 0,
// K % 4 == 0, M % 4 == 0 for transA / N % 4 == 0 for !transB) and a plain epilogue; the host picks it when that holds.
constexpr int TM = 128, TN = 128;

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm128_bf16_kernel(mmda_gemm_args g, int splitk, float* slab, int ldn) {
  __shared__ __attribute__((aligned(16))) unsigned short As[TM * LDS_BF16_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[TN * LDS_BF16_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int row0 = blockIdx.y * TM, col0 = blockIdx.x * TN;
  const int64_t bz = blockIdx.z / splitk;
  const int sp = blockIdx.z % splitk;
  const float* A = g.A + bz * g.strideA;
  const float* Bm = g.B + bz * g.strideB;
  float* C = g.C + bz * g.strideC;
  const int M = g.M, N = g.N, K = g.K;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gi = tid + 256 * i;
      float4 v = float4{0.f, 0.f, 0.f, 0.f};
      if (!TA) {                       // A (M,K): 8 groups of 4 k per row
        const int r = gi >> 3, k = k0 + (gi & 7) * 4;
        if (row0 + r < M && k < K) v = *reinterpret_cast<const float4*>(A + (int64_t)(row0 + r) * g.lda + k);
      } else {                         // A (K,M): 32 groups of 4 m per k-row
        const int k = k0 + (gi >> 5), m = row0 + (gi & 31) * 4;
        if (k < K && m < M) v = *reinterpret_cast<const float4*>(A + (int64_t)k * g.lda + m);
      }
      ra[i] = v;
      float4 u = float4{0.f, 0.f, 0.f, 0.f};
      if (TB) {                        // B (N,K)
        const int r = gi >> 3, k = k0 + (gi & 7) * 4;
        if (col0 + r < N && k < K) u = *reinterpret_cast<const float4*>(Bm + (int64_t)(col0 + r) * g.ldb + k);
      } else {                         // B (K,N)
        const int k = k0 + (gi >> 5), n = col0 + (gi & 31) * 4;
        if (k < K && n < N) u = *reinterpret_cast<const float4*>(Bm + (int64_t)k * g.ldb + n);
      }
      rb[i] = u;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gi = tid + 256 * i;
      if (!TA) {
        const int r = gi >> 3, k = (gi & 7) * 4;
        uint2 p; p.x = f2bf(ra[i].x) | ((unsigned)f2bf(ra[i].y) << 16); p.y = f2bf(ra[i].z) | ((unsigned)f2bf(ra[i].w) << 16);
        *reinterpret_cast<uint2*>(&As[r * LDS_BF16_LD + k]) = p;
      } else {
        const int k = gi >> 5, m = (gi & 31) * 4;
        As[(m + 0) * LDS_BF16_LD + k] = f2bf(ra[i].x); As[(m + 1) * LDS_BF16_LD + k] = f2bf(ra[i].y);
        As[(m + 2) * LDS_BF16_LD + k] = f2bf(ra[i].z); As[(m + 3) * LDS_BF16_LD + k] = f2bf(ra[i].w);
      }
      if (TB) {
        const int r = gi >> 3, k = (gi & 7) * 4;
        uint2 p; p.x = f2bf(rb[i].x) | ((unsigned)f2bf(rb[i].y) << 16); p.y = f2bf(rb[i].z) | ((unsigned)f2bf(rb[i].w) << 16);
        *reinterpret_cast<uint2*>(&Bs[r * LDS_BF16_LD + k]) = p;
      } else {
        const int k = gi >> 5, n = (gi & 31) * 4;
        Bs[(n + 0) * LDS_BF16_LD + k] = f2bf(rb[i].x); Bs[(n + 1) * LDS_BF16_LD + k] = f2bf(rb[i].y);
        Bs[(n + 2) * LDS_BF16_LD + k] = f2bf(rb[i].z); Bs[(n + 3) * LDS_BF16_LD + k] = f2bf(rb[i].w);
      }
    }
  };

  const int nk_all = (K + BK - 1) / BK;
  const int per = (nk_all + splitk - 1) / splitk;
  const int kt0 = sp * per;
  const int nk = min(nk_all, kt0 + per);
  if (splitk > 1 && kt0 >= nk) return;
  load_tile(kt0 * BK);
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = kt0; kt < nk; ++kt) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + i * 16 + fr) * LDS_BF16_LD + fq * 8]);
      b[i] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 64 + i * 16 + fr) * LDS_BF16_LD + fq * 8]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }

  const float* bias = g.bias ? g.bias + bz * g.strideBias : nullptr;
  const float* bias2 = g.bias2 ? g.bias2 + bz * g.strideBias : nullptr;
  const float alpha = g.alpha == 0.f ? 1.f : g.alpha;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col0 + wn * 64 + j * 16 + (lane & 15);
      if (n >= N) continue;
      float bsum = 0.f;
      if (bias) bsum += bias[n];
      if (bias2) bsum += bias2[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        const int64_t ci = (int64_t)m * g.ldc + n;
        if (splitk > 1) { slab[(((int64_t)bz * splitk + sp) * M + m) * ldn + n] = acc[i][j][r]; continue; }
        float v = alpha * acc[i][j][r] + bsum;
        if (g.accumulate) v += C[ci];
        C[ci] = v;
      }
    }
}

// column sums of X (M, N) added into out (and out2): stage 1 writes one partial row per row block, stage 2 adds the partials of a
// column in block order (no atomics: the same bits on every run)
__global__ void colsum_kernel(const float* __restrict__ X, int ld, int M, int N, float* part, int rows_per_block) {
  __shared__ float red[4][64];
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int rg = threadIdx.x >> 6;
  int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rg; r < r1; r += 4) s += X[(int64_t)r * ld + c];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < N) part[(int64_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void colsum_finish_kernel(const float* __restrict__ part, int nparts, int N, float* out, float* out2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float t = 0.f;
  for (int p = 0; p < nparts; ++p) t += part[(int64_t)p * N + c];
  out[c] += t;
  if (out2) out2[c] += t;
}

}  // namespace

extern "C" int mmda_gemm(const mmda_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) return MMDA_EINVAL;
  if (a->M < 0 || a->N < 0 || a->batch < 0 || a->K < 0 || (a->gather && a->transA)) return MMDA_EINVAL;
  if (a->M == 0 || a->N == 0 || a->batch == 0) return MMDA_OK;
  if (a->mode != MMDA_F32 && a->mode != MMDA_BF16) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // Split-K: these GEMMs are small (B=32: a few dozen output tiles) with long reductions (K = T*B for weight gradients,
  // 2048 for the FFN, 4H for the projections); one block per tile would leave 250 CUs idle behind a serial k-loop.
  const int Neff = a->N + (a->bias_grad ? 1 : 0);          // + the virtual ones column
  if (a->bias_grad && !a->transA) return MMDA_EINVAL;
  const int tiles = ceil_div(Neff, BN) * ceil_div(a->M, BM) * a->batch;
  const int nk = ceil_div(a->K, BK);
  int splitk = 1;
  const bool plain_epilogue = a->act == MMDA_ACT_NONE && a->drop_p <= 0.f && !a->gate;
  if (plain_epilogue && nk >= 8 && tiles <= 256) {
    splitk = (768 + tiles - 1) / tiles;
    int max_split = nk / 2;                       // >= 2 k-tiles (64 deep) per block
    if (splitk > max_split) splitk = max_split;
    if (splitk > 64) splitk = 64;
    if (splitk < 1) splitk = 1;
  }
  // no empty slices (an empty slice would leave its slab unwritten)
  auto norm_split = [](int nk_, int sk_) { const int per = ceil_div(nk_, sk_ < 1 ? 1 : sk_); return ceil_div(nk_, per); };
  splitk = norm_split(nk, splitk);
  auto reduce_job = [&](const float* slab, int ldn, int sk) {
    SplitKJob J = {};
    J.slab = slab; J.C = a->C; J.M = a->M; J.N = a->N; J.ldn = ldn; J.ldc = a->ldc; J.sk = sk; J.batch = a->batch;
    J.strideC = a->strideC; J.strideBias = a->strideBias; J.alpha = a->alpha; J.bias = a->bias; J.bias2 = a->bias2;
    J.bias_grad = a->bias_grad; J.bias_grad2 = a->bias_grad2; J.accumulate = a->accumulate;
    return J;
  };
  // large aligned bf16 GEMMs -> 128x128 tile kernel with 16-byte staging loads
  const bool aligned = ((a->lda | a->ldb | a->K) & 3) == 0 && (((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0 &&
                       ((a->strideA | a->strideB) & 3) == 0 && (!a->transA || (a->M & 3) == 0) && (a->transB || (a->N & 3) == 0);
  static const bool use128 = []() { const char* e = getenv("MMDA_GEMM128"); return !(e && e[0] == '0'); }();   // A/B switch (tools/)
  if (use128 && a->mode == MMDA_BF16 && plain_epilogue && aligned && !a->A2 && !a->gather && !a->bias_grad && a->M >= 128 && a->N >= 128) {
    const int tiles128 = ceil_div(a->N, TN) * ceil_div(a->M, TM) * a->batch;
    int sk = 1;
    if (nk >= 8 && tiles128 <= 256) {
      sk = (512 + tiles128 - 1) / tiles128;
      if (sk > nk / 2) sk = nk / 2;
      if (sk > 64) sk = 64;
      if (sk < 1) sk = 1;
    }
    sk = norm_split(nk, sk);
    float* slab = nullptr;
    const int ldn = a->N;
    if (sk > 1) {
      slab = mmda_scratch_get(s, sizeof(float) * (size_t)a->batch * sk * a->M * ldn);
      if (!slab) return MMDA_ELAUNCH;
    }
    dim3 grid128(ceil_div(a->N, TN), ceil_div(a->M, TM), a->batch * sk);
    if (grid128.y > 65535 || grid128.z > 65535) return MMDA_EINVAL;
    if (!a->transA && a->transB) hipLaunchKernelGGL((gemm128_bf16_kernel<false, true>), grid128, dim3(256), 0, s, *a, sk, slab, ldn);
    else if (!a->transA && !a->transB) hipLaunchKernelGGL((gemm128_bf16_kernel<false, false>), grid128, dim3(256), 0, s, *a, sk, slab, ldn);
    else if (a->transA && !a->transB) hipLaunchKernelGGL((gemm128_bf16_kernel<true, false>), grid128, dim3(256), 0, s, *a, sk, slab, ldn);
    else hipLaunchKernelGGL((gemm128_bf16_kernel<true, true>), grid128, dim3(256), 0, s, *a, sk, slab, ldn);
    MMDA_CHECK_LAUNCH("mmda_gemm(128)");
    if (sk > 1) { SplitKJob J = reduce_job(slab, ldn, sk); return mmda_splitk_reduce(&J, 1, s); }
    return MMDA_OK;
  }
  float* slab = nullptr;
  const int ldn = Neff;
  if (splitk > 1) {
    slab = mmda_scratch_get(s, sizeof(float) * (size_t)a->batch * splitk * a->M * ldn);
    if (!slab) return MMDA_ELAUNCH;
  }
  dim3 grid(ceil_div(Neff, BN), ceil_div(a->M, BM), a->batch * splitk);
  if (grid.y > 65535 || grid.z > 65535) return MMDA_EINVAL;
  const int per_split = ceil_div(nk, splitk);
  const bool tiny = per_split <= 4 && (int)(grid.x * grid.y * grid.z) <= 1024;
  if (a->mode == MMDA_BF16) {
    if (tiny) hipLaunchKernelGGL((gemm_kernel<MMDA_BF16, 4>), grid, dim3(256), 0, s, *a, splitk, slab, ldn);
    else hipLaunchKernelGGL((gemm_kernel<MMDA_BF16, 1>), grid, dim3(256), 0, s, *a, splitk, slab, ldn);
  } else {
    if (tiny) hipLaunchKernelGGL((gemm_kernel<MMDA_F32, 4>), grid, dim3(256), 0, s, *a, splitk, slab, ldn);
    else hipLaunchKernelGGL((gemm_kernel<MMDA_F32, 1>), grid, dim3(256), 0, s, *a, splitk, slab, ldn);
  }
  MMDA_CHECK_LAUNCH("mmda_gemm");
  if (splitk > 1) { SplitKJob J = reduce_job(slab, ldn, splitk); return mmda_splitk_reduce(&J, 1, s); }
  return MMDA_OK;
}

extern "C" int mmda_gemm_grouped(const mmda_gemm_args* args, int n, void* stream) {
  if (!args || n < 0) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n; base += GROUP_MAX) {
    GroupLaunch G;
    G.n = 0;
    int blocks = 0;
    const int cnt = (n - base) < GROUP_MAX ? (n - base) : GROUP_MAX;
    // size split-K so that the whole group offers ~4 workgroups per CU
    int tiles_total = 0;
    for (int i = 0; i < cnt; ++i) {
      const mmda_gemm_args& a = args[base + i];
      if (!a.A || !a.B || !a.C || a.M < 0 || a.N < 0 || a.K < 0 || a.batch < 0 || (a.gather && a.transA)) return MMDA_EINVAL;
      if (a.mode != MMDA_F32 && a.mode != MMDA_BF16) return MMDA_EINVAL;
      if (a.bias_grad && !a.transA) return MMDA_EINVAL;
      tiles_total += ceil_div(a.N + (a.bias_grad ? 1 : 0), BN) * ceil_div(a.M, BM) * a.batch;
    }
    std::vector<SplitKJob> jobs;
    std::vector<int> jk;                   // group slot of each job
    int64_t slab_floats = 0;
    for (int i = 0; i < cnt; ++i) {
      const mmda_gemm_args& a = args[base + i];
      if (a.M == 0 || a.N == 0 || a.batch == 0) continue;
      const int k = G.n++;
      G.p[k] = a;
      const int Ne = a.N + (a.bias_grad ? 1 : 0);
      G.tx[k] = ceil_div(Ne, BN); G.ty[k] = ceil_div(a.M, BM);
      const int nk = ceil_div(a.K, BK);
      int sk = 1;
      const bool plain = a.act == MMDA_ACT_NONE && a.drop_p <= 0.f && !a.gate;
      if (plain && nk >= 8 && tiles_total < 1024) {
        sk = (1024 + tiles_total - 1) / tiles_total;
        if (sk > nk / 2) sk = nk / 2;
        if (sk > 32) sk = 32;
        if (sk < 1) sk = 1;
      }
      { const int per = ceil_div(nk, sk); sk = ceil_div(nk, per); }     // no empty slices
      G.splitk[k] = sk;
      G.slab[k] = nullptr; G.ldn[k] = Ne;
      if (sk > 1) {
        SplitKJob J = {};
        J.C = a.C; J.M = a.M; J.N = a.N; J.ldn = Ne; J.ldc = a.ldc; J.sk = sk; J.batch = a.batch;
        J.strideC = a.strideC; J.strideBias = a.strideBias; J.alpha = a.alpha; J.bias = a.bias; J.bias2 = a.bias2;
        J.bias_grad = a.bias_grad; J.bias_grad2 = a.bias_grad2; J.accumulate = a.accumulate;
        J.slab = reinterpret_cast<const float*>((uintptr_t)slab_floats);      // offset for now; the base is added below
        slab_floats += (int64_t)a.batch * sk * a.M * Ne;
        jobs.push_back(J); jk.push_back(k);
      }
      G.start[k] = blocks;
      blocks += G.tx[k] * G.ty[k] * a.batch * sk;
    }
    for (int k = G.n; k <= GROUP_MAX; ++k) G.start[k] = blocks;
    for (int k = G.n; k < GROUP_MAX; ++k) { G.p[k] = G.p[0]; G.tx[k] = G.ty[k] = G.splitk[k] = 1; G.slab[k] = nullptr; G.ldn[k] = 0; }
    if (blocks == 0) continue;
    if (!jobs.empty()) {
      float* slab_base = mmda_scratch_get(s, sizeof(float) * (size_t)slab_floats);
      if (!slab_base) return MMDA_ELAUNCH;
      for (size_t j = 0; j < jobs.size(); ++j) {
        float* p = slab_base + (int64_t)(uintptr_t)jobs[j].slab;
        jobs[j].slab = p; G.slab[jk[j]] = p;
      }
    }
    hipLaunchKernelGGL(gemm_grouped_kernel, dim3(blocks), dim3(256), 0, s, G);
    MMDA_CHECK_LAUNCH("mmda_gemm_grouped");
    if (!jobs.empty()) { const int rc = mmda_splitk_reduce(jobs.data(), (int)jobs.size(), s); if (rc) return rc; }
  }
  return MMDA_OK;
}

extern "C" int mmda_colsum(const float* X, int ld, int M, int N, float* out, float* out2, void* stream) {
  if (!X || !out || M < 0 || N <= 0) return MMDA_EINVAL;
  if (M == 0) return MMDA_OK;
  int rpb = 256;
  dim3 grid(ceil_div(N, 64), ceil_div(M, rpb));
  float* part = mmda_scratch_get((hipStream_t)stream, sizeof(float) * (size_t)grid.y * N);
  if (!part) return MMDA_ELAUNCH;
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, ld, M, N, part, rpb);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, part, (int)grid.y, N, out, out2);
  MMDA_CHECK_LAUNCH("mmda_colsum");
  return MMDA_OK;
}
