// bf16-operand GEMM for the LSTM-sized products of the bf16 mode, and the fp32 -> bf16 (plain / transposed) conversion that
// feeds it.  The generic kernel in gemm.hip stages fp32 operands through registers and converts on the LDS write: that
// path is VALU-bound (16 scalar loads + 16 conversions + 16 ds_write_b16 per thread and k-tile against 4 MFMAs per wave) and
// tops out at 50-100 TFLOP/s.  Here both operands are bf16 and K-major in HBM, so a k-tile is 16-byte loads straight into
// 16-byte LDS stores, and every GEMM of the path (NT forward, NN input gradient, TN weight gradient) is brought to the one
// NT form by giving it the right (plain or transposed) copy.
//
// Tile 128 x 128 x 64, 256 threads = 4 waves (2 x 2), 64 x 64 per wave = 4 x 4 MFMA 16x16x32 accumulators x 2 k-steps:
// 32 MFMAs per wave and k-tile against 16 ds_read_b128 and 4+4 16-byte global loads per thread.  Two register stages
// of prefetch (k+1 and k+2) because these problems only offer ~1 workgroup per CU and cannot hide HBM latency by occupancy.
#include "common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int LDT = TK + 8;                 // bf16 elements per LDS row (144 B: 16-B aligned, conflict-light for ds_read_b128)
constexpr int GROUP_MAX = 16;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Bf16Group {
  mmda_gemm_bf16_args p[GROUP_MAX];
  int start[GROUP_MAX + 1];
  int tx[GROUP_MAX], ty[GROUP_MAX], splitk[GROUP_MAX];
  int n;
};

__device__ __forceinline__ u32x4 ld_chunk(const unsigned short* base, int row, int nrows, int ld, int k, int Kp) {
  // 8 bf16 = 16 B; rows past the matrix and k past the (8-padded) depth read as zero
  if (row < nrows && k < Kp) return *reinterpret_cast<const u32x4*>(base + (int64_t)row * ld + k);
  return u32x4{0u, 0u, 0u, 0u};
}

__global__ __launch_bounds__(256) void gemm_bf16_kernel(Bf16Group G) {
  __shared__ __attribute__((aligned(16))) unsigned short As[TM * LDT];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[TN * LDT];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < G.n && (int)blockIdx.x >= G.start[k]) pi = k;
  const mmda_gemm_bf16_args g = G.p[pi];
  const int splitk = G.splitk[pi];
  const int local = blockIdx.x - G.start[pi];
  const int bx = local % G.tx[pi], by = (local / G.tx[pi]) % G.ty[pi], sp = local / (G.tx[pi] * G.ty[pi]);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int row0 = by * TM, col0 = bx * TN;
  const int M = g.M, N = g.N, K = g.K;
  const int Kp = (K + 7) & ~7;
  const unsigned short* A = reinterpret_cast<const unsigned short*>(g.A);
  const unsigned short* Bm = reinterpret_cast<const unsigned short*>(g.B);
  const bool ones_row = g.bias_grad != nullptr;        // virtual all-ones row n == N of B: its output column is sum_k A[m,k]

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: 128 rows x 64 k = 1024 chunks of 8 bf16 per operand -> 4 per thread; chunk c: row = c >> 3, k = (c & 7) * 8
  u32x4 ra[2][4], rb[2][4];
  auto load_tile = [&](u32x4 (&a)[4], u32x4 (&b)[4], int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      const int r = c >> 3, k = k0 + (c & 7) * 8;
      a[i] = ld_chunk(A, row0 + r, M, g.lda, k, Kp);
      u32x4 v = ld_chunk(Bm, col0 + r, N, g.ldb, k, Kp);
      if (ones_row && col0 + r == N) {
        // bf16 1.0 = 0x3F80; elements past K stay zero so the sum runs over the real depth only
        unsigned e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = (k + q < K) ? 0x3F80u : 0u;
        v = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
      }
      b[i] = v;
    }
  };
  auto store_tile = [&](const u32x4 (&a)[4], const u32x4 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      const int r = c >> 3, k = (c & 7) * 8;
      *reinterpret_cast<u32x4*>(&As[r * LDT + k]) = a[i];
      *reinterpret_cast<u32x4*>(&Bs[r * LDT + k]) = b[i];
    }
  };

  const int nk_all = (K + TK - 1) / TK;
  const int per = (nk_all + splitk - 1) / splitk;
  const int kt0 = sp * per;
  const int nk = min(nk_all, kt0 + per);
  if (kt0 >= nk) return;
  const int fr = lane & 15, fq = lane >> 4;

  load_tile(ra[0], rb[0], kt0 * TK);
  if (kt0 + 1 < nk) load_tile(ra[1], rb[1], (kt0 + 1) * TK);
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + i * 16 + fr) * LDT + ks * 32 + fq * 8]);
        b[i] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 64 + i * 16 + fr) * LDT + ks * 32 + fq * 8]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  // unrolled by two so that the two register stages are addressed statically
  for (int kt = kt0; kt < nk; kt += 2) {
    __syncthreads();
    store_tile(ra[0], rb[0]);
    __syncthreads();
    if (kt + 2 < nk) load_tile(ra[0], rb[0], (kt + 2) * TK);
    compute();
    if (kt + 1 < nk) {
      __syncthreads();
      store_tile(ra[1], rb[1]);
      __syncthreads();
      if (kt + 3 < nk) load_tile(ra[1], rb[1], (kt + 3) * TK);
      compute();
    }
  }

  const float alpha = g.alpha == 0.f ? 1.f : g.alpha;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col0 + wn * 64 + j * 16 + (lane & 15);
      if (ones_row && n == N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
          if (m < M) {
            atomicAdd(&g.bias_grad[m], acc[i][j][r]);
            if (g.bias_grad2) atomicAdd(&g.bias_grad2[m], acc[i][j][r]);
          }
        }
        continue;
      }
      if (n >= N) continue;
      float bsum = 0.f;
      if (g.bias) bsum += g.bias[n];
      if (g.bias2) bsum += g.bias2[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        const int64_t ci = (int64_t)m * g.ldc + n;
        if (splitk > 1) { atomicAdd(&g.C[ci], alpha * acc[i][j][r] + (sp == 0 ? bsum : 0.f)); continue; }
        float v = alpha * acc[i][j][r] + bsum;
        if (g.accumulate) v += g.C[ci];
        g.C[ci] = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------ conversion
// 64 x 64 tiles through LDS: coalesced fp32 reads along the source rows, coalesced bf16 writes along the rows of the plain
// copy and (transposed through LDS) along the rows of the transposed copy.  Padding columns up to the leading dimension are
// written as zeros so that the GEMM can read whole 16-byte chunks.
struct ConvLaunch {
  mmda_convert_job j[GROUP_MAX];
  int start[GROUP_MAX + 1];
  int tx[GROUP_MAX];
  int n;
};

__global__ __launch_bounds__(256) void convert_kernel(ConvLaunch L) {
  __shared__ unsigned short tile[64][66];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_convert_job J = L.j[pi];
  const int local = blockIdx.x - L.start[pi];
  const int bx = local % L.tx[pi], by = local / L.tx[pi];
  const int r0 = by * 64, c0 = bx * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  unsigned short* P = reinterpret_cast<unsigned short*>(J.plain);
  unsigned short* Tt = reinterpret_cast<unsigned short*>(J.transposed);
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    float v = 0.f;
    if (r < J.rows && c < J.cols) {
      const int64_t sr = J.gather ? J.gather[r] : r;
      v = J.src[sr * J.ld + c];
    }
    const unsigned short h = f2bf(v);
    tile[rr][tx] = h;
    if (P && r < J.rows && c < J.ldp) P[(int64_t)r * J.ldp + c] = h;      // c in [cols, ldp) writes the zero padding
  }
  if (!Tt) return;
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, r = r0 + tx;
    if (c < J.cols && r < J.ldt) Tt[(int64_t)c * J.ldt + r] = tile[tx][cc];   // r in [rows, ldt) carries zeros (loaded as 0 above)
  }
}

}  // namespace

extern "C" int mmda_gemm_bf16_grouped(const mmda_gemm_bf16_args* args, int n, void* stream) {
  if (!args || n < 0) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n; base += GROUP_MAX) {
    const int cnt = (n - base) < GROUP_MAX ? (n - base) : GROUP_MAX;
    Bf16Group G;
    G.n = 0;
    int tiles_total = 0;
    for (int i = 0; i < cnt; ++i) {
      const mmda_gemm_bf16_args& a = args[base + i];
      if (!a.A || !a.B || !a.C || a.M < 0 || a.N < 0 || a.K < 0) return MMDA_EINVAL;
      if ((a.lda & 7) || (a.ldb & 7) || (((uintptr_t)a.A | (uintptr_t)a.B) & 15)) return MMDA_EINVAL;
      if (a.lda < ((a.K + 7) & ~7) || a.ldb < ((a.K + 7) & ~7)) return MMDA_EINVAL;
      tiles_total += ceil_div(a.N + (a.bias_grad ? 1 : 0), TN) * ceil_div(a.M, TM);
    }
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      const mmda_gemm_bf16_args& a = args[base + i];
      if (a.M == 0 || a.N == 0) continue;
      const int k = G.n++;
      G.p[k] = a;
      G.tx[k] = ceil_div(a.N + (a.bias_grad ? 1 : 0), TN); G.ty[k] = ceil_div(a.M, TM);
      const int nk = ceil_div(a.K, TK);
      int sk = 1;
      if (nk >= 4 && tiles_total < 512) {            // ~2 workgroups per CU over the whole group
        sk = (512 + tiles_total - 1) / tiles_total;
        if (sk > nk / 2) sk = nk / 2;
        if (sk > 32) sk = 32;
        if (sk < 1) sk = 1;
      }
      if (sk > 1 && !a.accumulate) {
        if (a.ldc != a.N) sk = 1;
        else if (hipMemsetAsync(a.C, 0, sizeof(float) * (size_t)a.M * a.N, s) != hipSuccess) return MMDA_ELAUNCH;
      }
      G.splitk[k] = sk;
      G.start[k] = blocks;
      blocks += G.tx[k] * G.ty[k] * sk;
    }
    for (int k = G.n; k <= GROUP_MAX; ++k) G.start[k] = blocks;
    for (int k = G.n; k < GROUP_MAX; ++k) { G.p[k] = G.p[0]; G.tx[k] = G.ty[k] = G.splitk[k] = 1; }
    if (blocks == 0) continue;
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3(blocks), dim3(256), 0, s, G);
    MMDA_CHECK_LAUNCH("mmda_gemm_bf16_grouped");
  }
  return MMDA_OK;
}

extern "C" int mmda_convert_bf16(const mmda_convert_job* jobs, int n, void* stream) {
  if (!jobs || n < 0) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n; base += GROUP_MAX) {
    const int cnt = (n - base) < GROUP_MAX ? (n - base) : GROUP_MAX;
    ConvLaunch L;
    L.n = 0;
    int blocks = 0;
    for (int i = 0; i < cnt; ++i) {
      const mmda_convert_job& j = jobs[base + i];
      if (!j.src || j.rows < 0 || j.cols < 0 || (!j.plain && !j.transposed)) return MMDA_EINVAL;
      if (j.plain && (j.ldp < ((j.cols + 7) & ~7) || (j.ldp & 7))) return MMDA_EINVAL;
      if (j.transposed && (j.ldt < ((j.rows + 7) & ~7) || (j.ldt & 7))) return MMDA_EINVAL;
      if (j.rows == 0 || j.cols == 0) continue;
      const int k = L.n++;
      L.j[k] = j;
      // tiles cover the padded extents so that the zero padding gets written
      const int ext_c = j.plain ? (j.ldp > j.cols ? j.ldp : j.cols) : j.cols;
      const int ext_r = j.transposed ? (j.ldt > j.rows ? j.ldt : j.rows) : j.rows;
      L.tx[k] = ceil_div(ext_c, 64);
      L.start[k] = blocks;
      blocks += L.tx[k] * ceil_div(ext_r, 64);
    }
    for (int k = L.n; k <= GROUP_MAX; ++k) L.start[k] = blocks;
    for (int k = L.n; k < GROUP_MAX; ++k) { L.j[k] = L.j[0]; L.tx[k] = 1; }
    if (blocks == 0) continue;
    hipLaunchKernelGGL(convert_kernel, dim3(blocks), dim3(256), 0, s, L);
    MMDA_CHECK_LAUNCH("mmda_convert_bf16");
  }
  return MMDA_OK;
}
