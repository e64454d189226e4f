// Row-skinny exact-f32 GEMM for the fusion block (projections, private/shared/recon, transformer layer, heads and their
// input gradients): M = B or 6B rows (32..192 at the reference batch size), N and K between 12 and 2048.  At these sizes
// the generic 64x64 tile kernel runs on two to a few dozen workgroups and is bound by its serial k-loop; the chain of
// ~45 such GEMMs was a quarter of the training step.  Here
//   * one workgroup owns a 32 x 16 output tile and its EIGHT waves split K (interleaved 16-deep chunks), so even
//     M=32,N=128 gives 8 workgroups x 8 waves and K=1200 is 10 chunks per wave;
//   * operands go global -> registers -> MFMA (v_mfma_f32_16x16x4_f32, exact) with no LDS staging: every weight element
//     is used by exactly one workgroup column, so LDS would only add a round trip.  The 16x16x4 fragment wants, per lane,
//     A[row = lane&15][k = lane>>4]; a lane loads the float4 A[row][16c + 4(lane>>4) .. +3] and feeds component s to the
//     s-th of four MFMAs -- a permutation of k inside the chunk that A and B share, so the sum is unchanged;
//   * four chunks of loads are in flight per wave before the first MFMA (the whole K range for most problems);
//   * the eight partial tiles meet in LDS (16 KB), one output element per thread, then the fused epilogue:
//     alpha, bias, accumulate, activation, dropout, relu-gate, sigmoid-backward factor, second destination.
// Up to 8 independent problems per launch.  Deterministic: no atomics, fixed reduction order.
#include "common.h"
#include "transpose_tile.h"
#include <stdlib.h>

namespace {

constexpr int SK_WAVES = 8, SK_MAXP = 8, SK_PF = 4;
constexpr int SK_TM = 32, SK_TN = 16;

struct SkinnyLaunch {
  mmda_skinny_args p[SK_MAXP];
  int start[SK_MAXP + 1];
  int tx[SK_MAXP];
  int nsplit[SK_MAXP];           // 1: waves split N (short K), 0: waves split K
  int n;
};

typedef __attribute__((ext_vector_type(4))) float f4;
__device__ __forceinline__ bool vec_ok(const float* p, int ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// One product: acc[t] += A[row0 + 16t .. , :] * op(B)[:, col0 ..] over this wave's chunks.
// VEC: float4 loads along k for A (and for B when TB); needs 16-B aligned bases, ld % 4 == 0 and K % 4 == 0.
template <bool TB, bool VEC, int WK>
__device__ __forceinline__ void skinny_product(f32x4 (&acc)[2], const float* __restrict__ A, const float* __restrict__ A2, int lda,
                                               const float* __restrict__ Bm, int ldb, int M, int N, int K, int row0, int col0,
                                               int wave, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const int nchunks = (K + 15) >> 4;
  // clamped (always valid) addresses; out-of-range rows / columns / k are zeroed after the load so that no load sits under
  // a lane-dependent branch
  const int ar0 = min(row0 + r, M - 1), ar1 = min(row0 + 16 + r, M - 1);
  const bool a0_ok = row0 + r < M, a1_ok = row0 + 16 + r < M;
  const int bc = min(col0 + r, N - 1);
  const bool b_ok = col0 + r < N;
  f4 ra0[SK_PF], ra1[SK_PF], rb[SK_PF];

  auto load = [&](int slot, int c) {
    const int k = c * 16 + 4 * g;
    if (VEC) {
      const int kc = min(k, K - 4);
      const bool k_ok = k < K;
      f4 a0 = *reinterpret_cast<const f4*>(A + (int64_t)ar0 * lda + kc);
      f4 a1 = *reinterpret_cast<const f4*>(A + (int64_t)ar1 * lda + kc);
      if (A2) {
        a0 += *reinterpret_cast<const f4*>(A2 + (int64_t)ar0 * lda + kc);
        a1 += *reinterpret_cast<const f4*>(A2 + (int64_t)ar1 * lda + kc);
      }
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      ra0[slot] = (a0_ok && k_ok) ? a0 : z;
      ra1[slot] = (a1_ok && k_ok) ? a1 : z;
      if (TB) {
        f4 b = *reinterpret_cast<const f4*>(Bm + (int64_t)bc * ldb + kc);
        rb[slot] = (b_ok && k_ok) ? b : z;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kc = min(k + s, K - 1);
        const bool k_ok = k + s < K;
        float a0 = A[(int64_t)ar0 * lda + kc], a1 = A[(int64_t)ar1 * lda + kc];
        if (A2) { a0 += A2[(int64_t)ar0 * lda + kc]; a1 += A2[(int64_t)ar1 * lda + kc]; }
        ra0[slot][s] = (a0_ok && k_ok) ? a0 : 0.f;
        ra1[slot][s] = (a1_ok && k_ok) ? a1 : 0.f;
        if (TB) {
          float b = Bm[(int64_t)bc * ldb + kc];
          rb[slot][s] = (b_ok && k_ok) ? b : 0.f;
        }
      }
    }
    if (!TB) {
      // B is (K, ldb) with n contiguous: four rows k..k+3, 64 B per 16 lanes each
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kc = min(k + s, K - 1);
        float b = Bm[(int64_t)kc * ldb + bc];
        rb[slot][s] = (b_ok && k + s < K) ? b : 0.f;
      }
    }
  };
  auto mma = [&](int slot) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra0[slot][s], rb[slot][s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[slot][s], rb[slot][s], acc[1], 0, 0, 0);
    }
  };

  if (wave >= nchunks) return;                       // wave-uniform
#pragma unroll
  for (int p = 0; p < SK_PF; ++p) load(p, wave + WK * p);            // chunks past the end load clamped addresses and read as zero
  for (int base = wave; base < nchunks; base += WK * SK_PF) {
#pragma unroll
    for (int p = 0; p < SK_PF; ++p) {
      if (base + WK * p < nchunks) mma(p);           // wave-uniform
      const int next = base + WK * (p + SK_PF);
      if (next < nchunks) load(p, next);             // wave-uniform
    }
  }
}

// the fused epilogue of one output element
__device__ __forceinline__ void skinny_epilogue(const mmda_skinny_args& g, int m, int n, float raw) {
  raw *= (g.alpha == 0.f ? 1.f : g.alpha);
  float v = raw;
  if (g.bias) v += g.bias[n];
  const int64_t ci = (int64_t)m * g.ldc + n;
  if (g.accumulate) v += g.C[ci];
  v = act_fwd(g.act, v);
  if (g.drop_p > 0.f) v *= drop_mul(g.drop_p, g.drop_seed, g.drop_site, (uint64_t)m * g.N + n);
  if (g.gate) v *= (g.gate[(int64_t)m * g.ldgate + n] > 0.f) ? g.gate_scale : 0.f;
  if (g.dsig) { const float s = g.dsig[(int64_t)m * g.lddsig + n]; v *= s * (1.f - s); }
  g.C[ci] = v;
  if (g.C2) {
    // second destination of the same product (the gradient of private + shared flows into both halves of d_x6)
    const int64_t c2 = (int64_t)m * g.ldc2 + n;
    float u = raw;
    if (g.accumulate) u += g.C2[c2];
    if (g.dsig2) { const float s = g.dsig2[(int64_t)m * g.lddsig + n]; u *= s * (1.f - s); }
    g.C2[c2] = u;
  }
}

// Short K, wide N (the FFN up-projection and its input gradient: M = 6B, N = 2048, K = 128): no K split.  The block's eight
// waves tile a 64 x 128 output (2 x 4 waves of 32 x 32 = 2 x 2 MFMA tiles each), every wave walks the whole K with four
// float4 loads (two A row tiles, two B column tiles) per 16-deep chunk feeding 16 MFMAs, and writes its tile straight from
// the accumulators.  The K-split form spends 768 workgroups on one chunk each for this shape.  NT, 16-byte aligned operands only.
__device__ __forceinline__ void skinny_wide(const mmda_skinny_args& g, int row0, int col0, int lane) {
  const int r = lane & 15, q = lane >> 4;
  const int M = g.M, N = g.N, K = g.K;
  const int nchunks = (K + 15) >> 4;
  const int ar0 = min(row0 + r, M - 1), ar1 = min(row0 + 16 + r, M - 1);
  const int bc0 = min(col0 + r, N - 1), bc1 = min(col0 + 16 + r, N - 1);
  const bool a0_ok = row0 + r < M, a1_ok = row0 + 16 + r < M, b0_ok = col0 + r < N, b1_ok = col0 + 16 + r < N;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int PF = 4;
  f4 ra0[PF], ra1[PF], rb0[PF], rb1[PF];
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int slot, int c) {
    const int k = c * 16 + 4 * q;
    const int kc = min(k, K - 4);
    const bool k_ok = k < K;
    f4 a0 = *reinterpret_cast<const f4*>(g.A + (int64_t)ar0 * g.lda + kc);
    f4 a1 = *reinterpret_cast<const f4*>(g.A + (int64_t)ar1 * g.lda + kc);
    if (g.A2) {
      a0 += *reinterpret_cast<const f4*>(g.A2 + (int64_t)ar0 * g.lda + kc);
      a1 += *reinterpret_cast<const f4*>(g.A2 + (int64_t)ar1 * g.lda + kc);
    }
    const f4 b0 = *reinterpret_cast<const f4*>(g.B + (int64_t)bc0 * g.ldb + kc);
    const f4 b1 = *reinterpret_cast<const f4*>(g.B + (int64_t)bc1 * g.ldb + kc);
    ra0[slot] = (a0_ok && k_ok) ? a0 : z; ra1[slot] = (a1_ok && k_ok) ? a1 : z;
    rb0[slot] = (b0_ok && k_ok) ? b0 : z; rb1[slot] = (b1_ok && k_ok) ? b1 : z;
  };
  auto mma = [&](int slot) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra0[slot][s], rb0[slot][s], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra0[slot][s], rb1[slot][s], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[slot][s], rb0[slot][s], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[slot][s], rb1[slot][s], acc[1][1], 0, 0, 0);
    }
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) load(p, p);
  for (int base = 0; base < nchunks; base += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      if (base + p < nchunks) mma(p);
      if (base + p + PF < nchunks) load(p, base + p + PF);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = col0 + j * 16 + (lane & 15);
      if (n >= N) continue;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int m = row0 + i * 16 + (lane >> 4) * 4 + rg;
        if (m < M) skinny_epilogue(g, m, n, acc[i][j][rg]);
      }
    }
}

template <int WK>
__device__ __forceinline__ void skinny_products(f32x4 (&acc)[2], const mmda_skinny_args& g, int row0, int col0, int wave, int lane) {
  for (int prod = 0; prod < 2; ++prod) {
    const float* A = prod ? g.A_2nd : g.A;
    const float* A2 = prod ? nullptr : g.A2;
    const float* Bm = prod ? g.B_2nd : g.B;
    const int lda = prod ? g.lda_2nd : g.lda, ldb = prod ? g.ldb_2nd : g.ldb, K = prod ? g.K2 : g.K;
    if (!A || K <= 0) continue;
    const bool va = vec_ok(A, lda) && (K & 3) == 0 && (!A2 || vec_ok(A2, lda));
    if (g.transB) {
      if (va && vec_ok(Bm, ldb)) skinny_product<true, true, WK>(acc, A, A2, lda, Bm, ldb, g.M, g.N, K, row0, col0, wave, lane);
      else skinny_product<true, false, WK>(acc, A, A2, lda, Bm, ldb, g.M, g.N, K, row0, col0, wave, lane);
    } else {
      if (va) skinny_product<false, true, WK>(acc, A, A2, lda, Bm, ldb, g.M, g.N, K, row0, col0, wave, lane);
      else skinny_product<false, false, WK>(acc, A, A2, lda, Bm, ldb, g.M, g.N, K, row0, col0, wave, lane);
    }
  }
}

__global__ __launch_bounds__(512) void gemm_skinny_kernel(SkinnyLaunch L) {
  __shared__ float red[SK_WAVES][2][256];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < SK_MAXP; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_skinny_args& g = L.p[pi];
  const int local = blockIdx.x - L.start[pi];
  const int bx = local % L.tx[pi], by = local / L.tx[pi];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (L.nsplit[pi] == 2) {
    // 64 x 128 block tile, wave (wr, wc) = (wave >> 2, wave & 3) owns 32 x 32
    const int row0 = by * 64 + (wave >> 2) * 32, col0 = bx * 128 + (wave & 3) * 32;
    if (row0 >= g.M || col0 >= g.N) return;              // wave-uniform; no barrier on this path
    skinny_wide(g, row0, col0, lane);
    return;
  }
  if (L.nsplit[pi]) {
    // short K, wide N: the eight waves take eight adjacent 16-column tiles and each walks the whole K; no cross-wave sum
    const int row0 = by * SK_TM, col0 = (bx * SK_WAVES + wave) * SK_TN;
    if (col0 >= g.N) return;                             // wave-uniform; no barrier on this path
    skinny_products<1>(acc, g, row0, col0, 0, lane);
    const int n = col0 + (lane & 15);
    if (n >= g.N) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int m = row0 + t * 16 + (lane >> 4) * 4 + rg;
        if (m < g.M) skinny_epilogue(g, m, n, acc[t][rg]);
      }
    return;
  }
  const int row0 = by * SK_TM, col0 = bx * SK_TN;
  skinny_products<SK_WAVES>(acc, g, row0, col0, wave, lane);
  // D fragment: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) red[wave][t][((lane >> 4) * 4 + rg) * 16 + (lane & 15)] = acc[t][rg];
  __syncthreads();
  const int t = tid >> 8, e = tid & 255;
  const int m = row0 + t * 16 + (e >> 4), n = col0 + (e & 15);
  if (m >= g.M || n >= g.N) return;
  float raw = 0.f;
#pragma unroll
  for (int w = 0; w < SK_WAVES; ++w) raw += red[w][t][e];      // fixed order: bitwise reproducible
  skinny_epilogue(g, m, n, raw);
}

// fp32 transposes (up to 20 per launch, transpose_tile.h): dst[c][r] = src[r][c] through 32 x 33 LDS tiles, coalesced on both sides.  Used once per
// step (side stream) for the K-major copies of the fusion block's weights that its input-gradient GEMMs read.
__global__ __launch_bounds__(256) void transpose_kernel(TrLaunch L) {
  __shared__ float tile[32][33];
  transpose_block(L, (int)blockIdx.x, tile);
}

}  // namespace

extern "C" int mmda_gemm_skinny(const mmda_skinny_args* args, int n, void* stream) {
  if (!args || n < 0) return MMDA_EINVAL;
  for (int i = 0; i < n; ++i) {
    const mmda_skinny_args& a = args[i];
    if (!a.A || !a.B || !a.C || a.M < 0 || a.N < 0 || a.K <= 0 || a.lda < a.K || a.ldc < a.N) return MMDA_EINVAL;
    if (a.transB ? a.ldb < a.K : a.ldb < a.N) return MMDA_EINVAL;
    if (a.K2 > 0 && (!a.A_2nd || !a.B_2nd || a.lda_2nd < a.K2 || (a.transB ? a.ldb_2nd < a.K2 : a.ldb_2nd < a.N))) return MMDA_EINVAL;
    if (a.C2 && (a.ldc2 < a.N || a.bias || a.act || a.drop_p > 0.f || a.gate)) return MMDA_EINVAL;   // C2 carries alpha/accumulate/dsig2 only
  }
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n; base += SK_MAXP) {
    const int cnt = (n - base) < SK_MAXP ? (n - base) : SK_MAXP;
    SkinnyLaunch L;
    L.n = 0;
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      const mmda_skinny_args& a = args[base + i];
      if (a.M == 0 || a.N == 0) continue;
      const int k = L.n++;
      L.p[k] = a;
      // Column-split form (one wave per 16-column tile walks the whole K; K <= 256): OFF unless MMDA_SKINNY_NSPLIT_MIN_N is set.
      // Measured on the training step it loses to the K-split form at every shape of the fusion block (1.35 -> 1.43 ms per
      // step with it on for N >= 128, 1.37 for N >= 1024): eight waves sharing one k-walk's latency beat eight walking alone.
      static const int nsplit_min_n = getenv("MMDA_SKINNY_NSPLIT_MIN_N") ? atoi(getenv("MMDA_SKINNY_NSPLIT_MIN_N")) : (1 << 30);
      L.nsplit[k] = (a.K <= 256 && a.K2 <= 256 && a.N >= nsplit_min_n) ? 1 : 0;
      // wide form: NT, single product, K <= 256 and a multiple of 4, N >= 1024, 16-byte aligned operands
      static const int wide_min_n = getenv("MMDA_SKINNY_WIDE_MIN_N") ? atoi(getenv("MMDA_SKINNY_WIDE_MIN_N")) : (1 << 30);    // OFF by default: measured +15 us per step against the K-split form at N = 2048
      const bool al = ((((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.A2) & 15) == 0) && !(a.lda & 3) && !(a.ldb & 3) && !(a.K & 3);
      if (a.transB && a.K2 <= 0 && a.K <= 256 && a.N >= wide_min_n && al && !a.C2) L.nsplit[k] = 2;
      int rows_per_block = SK_TM;
      if (L.nsplit[k] == 2) { L.tx[k] = ceil_div(a.N, 128); rows_per_block = 64; }
      else L.tx[k] = L.nsplit[k] ? ceil_div(a.N, SK_WAVES * SK_TN) : ceil_div(a.N, SK_TN);
      L.start[k] = blocks;
      blocks += L.tx[k] * ceil_div(a.M, rows_per_block);
    }
    for (int k = L.n; k <= SK_MAXP; ++k) L.start[k] = blocks;
    for (int k = L.n; k < SK_MAXP; ++k) { L.p[k] = L.p[0]; L.tx[k] = 1; L.nsplit[k] = 0; }
    if (blocks == 0) continue;
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3(blocks), dim3(512), 0, s, L);
    MMDA_CHECK_LAUNCH("mmda_gemm_skinny");
  }
  return MMDA_OK;
}

extern "C" int mmda_transpose_f32(const mmda_transpose_job* jobs, int n, void* stream) {
  if (!jobs || n < 0) return MMDA_EINVAL;
  for (int base = 0; base < n; base += TR_MAX) {
    TrLaunch L;
    int blocks = 0;
    const int rc = tr_build(jobs + base, (n - base) < TR_MAX ? (n - base) : TR_MAX, L, blocks);
    if (rc) return rc;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(transpose_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
    MMDA_CHECK_LAUNCH("mmda_transpose_f32");
  }
  return MMDA_OK;
}
