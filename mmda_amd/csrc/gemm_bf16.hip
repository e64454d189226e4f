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
#include "convert_tile.h"
#include "splitk.h"
#include <algorithm>
#include <stdlib.h>
#include <vector>

namespace {

constexpr int TK = 64;
constexpr int LDT = TK + 8;                 // bf16 elements per LDS row (144 B: 16-B aligned, conflict-light for ds_read_b128)
// tn form: the LDS image of an operand tile is [k 0..63][T + pad] (k-rows as they lie in memory); row strides of 40 (T = 64) and 72
// (T = 128) banks make eight consecutive k-rows of 16 columns tile the 64 banks exactly, which is what one half-wave of a transposing
// read touches (see tn_frag)
template <int T> struct TnLd { static constexpr int v = T == 64 ? 80 : 144; };
template <int T> struct LdsElems { static constexpr int nt = 2 * 2 * T * LDT, tn = 2 * 2 * TK * TnLd<T>::v, v = nt > tn ? nt : tn; };
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int GROUP_MAX = 16;


struct Bf16Group {
  mmda_gemm_bf16_args p[GROUP_MAX];
  int start[GROUP_MAX + 1];          // first block of each problem: multiples of 8 (see the XCD-aware tile order in gemm_bf16_kernel)
  int tx[GROUP_MAX], ty[GROUP_MAX], splitk[GROUP_MAX], tile[GROUP_MAX];
  float* slab[GROUP_MAX];            // split-K: [slice][M][ldn] raw partial tiles (splitk.h); null when splitk == 1
  int ldn[GROUP_MAX];
  int n;
};

__device__ __forceinline__ u32x4 ld_chunk(const unsigned short* base, int row, int nrows, int ld, int k, int Kp) {
  // 8 bf16 = 16 B; rows past the matrix and k past the (8-padded) depth read as zero.  The load itself is unconditional from a
  // clamped (always valid) address: a load under a lane-dependent branch costs a full s_waitcnt vmcnt(0) per chunk.
  // The zeroing happens when the chunk is STORED to LDS (store_tile), not here: a select on the loaded value right after the
  // load would make the compiler wait for it at once and the two-tile prefetch would be gone.
  const int rc = min(row, nrows - 1), kc = min(k, Kp - 8);
  return *reinterpret_cast<const u32x4*>(base + (int64_t)rc * ld + kc);
}

// Epilogue of one T x T output tile held as 16 x 16 accumulator fragments (acc[i][j]: rows wm * T/2 + 16 i + 4 (lane >> 4) + r,
// column wn * T/2 + 16 j + (lane & 15)).  `Cs`: LDS no longer read by anybody once every wave has passed the barrier inside (the operand
// buffers): the tile is staged there for 16-byte row stores.  Column n == N of a tile that holds it is the bias gradient.
template <int T>
__device__ __forceinline__ void gemm_bf16_epilogue(const mmda_gemm_bf16_args& g, f32x4 (&acc)[T / 32][T / 32], float* Cs, int row0, int col0,
                                                   int splitk, int sp, float* slab, int ldn, bool tile_has_ones) {
  constexpr int W = T / 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int M = g.M, N = g.N;
  const float alpha = g.alpha == 0.f ? 1.f : g.alpha;
  // ---- split-K: the raw partial tile (the bias-gradient column with it) goes into this slice's slab; the reduce launch behind the
  // GEMM sums the slabs in slice order and applies the epilogue (splitk.h) -- no float atomics, identical bits on every run
  const bool to_slab = splitk > 1;                       // block-uniform
  if (!to_slab && tile_has_ones && wn == (N - col0) / (T / 2)) {
    // the column n == N holds sum_k A[m,k]: bias gradient(s).  One tile per row block holds that column: a single writer per entry.
    const int j = ((N - col0) % (T / 2)) / 16;
    if ((lane & 15) == (N - col0) % 16) {
#pragma unroll
      for (int i = 0; i < W; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = row0 + wm * (T / 2) + i * 16 + (lane >> 4) * 4 + r;
          float v = 0.f;
#pragma unroll
          for (int jj = 0; jj < W; ++jj) v = (jj == j) ? acc[i][jj][r] : v;
          if (m < M) {
            const int mo = g.perm_m_H > 0 ? gate_orig(m, g.perm_m_H) : m;
            g.bias_grad[mo] += v;
            if (g.bias_grad2) g.bias_grad2[mo] += v;
          }
        }
    }
  }
  // ---- C tile.  With 16-byte-aligned rows the tile goes out through LDS: the accumulator fragments hold 4
  // rows x 16 columns per wave-register, so storing them directly is 64-byte pieces (a 1600 x 2400 fp32 output took 16 us
  // that way, 4x a memset of the same size); staged, every store instruction writes 16 bytes per lane along a row and the
  // bias / accumulate reads are 16-byte loads of the same shape.  (Slab rows are always aligned.)
  const bool vec_out = to_slab || ((g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0 && (N & 3) == 0);      // block-uniform
  if (vec_out) {
    constexpr int LDC = T + 4;                           // floats per staged row
    constexpr int RP = T == 128 ? 32 : 64;               // rows per pass: RP * LDC * 4 bytes <= the operand LDS block
    constexpr int C4 = T / 4;                            // float4 per staged row; divides 256, so a thread keeps ONE column group
        const int c4 = tid % C4;
    const int n = col0 + c4 * 4;
    const bool n_ok = n < (to_slab ? ldn : N);
    const float scale = to_slab ? 1.f : alpha;
    float* const out_base = to_slab ? slab + (int64_t)sp * M * ldn : g.C;
    const int out_ld = to_slab ? ldn : g.ldc;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    if (!to_slab && (g.bias || g.bias2)) {
      // n is a multiple of 4: with the gate interleave the four columns are the four gates of ONE unit, orig = nb0 + e * H
      const int nn = min(n, N - 4);
      int nb0 = nn, nbs = 1;
      if (g.perm_n_H > 0) { const int G = 4 * g.perm_n_H, d = nn / G; nb0 = d * G + ((nn - d * G) >> 2); nbs = g.perm_n_H; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (g.bias) bsum[e] += g.bias[nb0 + e * nbs];
        if (g.bias2) bsum[e] += g.bias2[nb0 + e * nbs];
      }
    }
#pragma unroll
    for (int p = 0; p < T / RP; ++p) {
      __syncthreads();                                   // operand tiles (first pass) / previous pass are no longer read
#pragma unroll
      for (int i = 0; i < W; ++i) {
        const int lr0 = wm * (T / 2) + i * 16 - p * RP;  // first row of this fragment block within the pass
        if (lr0 >= 0 && lr0 < RP) {                      // wave-uniform
#pragma unroll
          for (int j = 0; j < W; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Cs[(lr0 + (lane >> 4) * 4 + r) * LDC + wn * (T / 2) + j * 16 + (lane & 15)] = scale * acc[i][j][r];
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RP * C4 / 256; ++q) {
        const int lr = (q * 256 + tid) / C4;
        const int m = row0 + p * RP + lr;
        if (m < M && n_ok) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * LDC + c4 * 4]) + bsum;
          const int mo = (!to_slab && g.perm_m_H > 0) ? gate_orig(m, g.perm_m_H) : m;
          float* dst = out_base + (int64_t)mo * out_ld + n;
          if (!to_slab && g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = v;
        }
      }
    }
    return;
  }
  // Scalar path (unaligned rows).  Accumulating reads C first: all loads are issued from
  // clamped addresses before the first add (a load under the m < M / n < N branches would be waited for one by one).
#pragma unroll
  for (int i = 0; i < W; ++i) {
    float oldc[W][4];
    const bool rmw = g.accumulate;                             // block-uniform
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nc = min(col0 + wn * (T / 2) + j * 16 + (lane & 15), N - 1);
        int mc = min(row0 + wm * (T / 2) + i * 16 + (lane >> 4) * 4 + r, M - 1);
        if (g.perm_m_H > 0) mc = gate_orig(mc, g.perm_m_H);
        oldc[j][r] = rmw ? g.C[(int64_t)mc * g.ldc + nc] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int n = col0 + wn * (T / 2) + j * 16 + (lane & 15);
      const bool n_ok = n < N;
      const int nc = min(n, N - 1);
      float bsum = 0.f;
      const int nb = g.perm_n_H > 0 ? gate_orig(nc, g.perm_n_H) : nc;
      if (g.bias) bsum += g.bias[nb];
      if (g.bias2) bsum += g.bias2[nb];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = row0 + wm * (T / 2) + i * 16 + (lane >> 4) * 4 + r;
        if (!n_ok || m >= M) continue;
        const int64_t ci = (int64_t)(g.perm_m_H > 0 ? gate_orig(m, g.perm_m_H) : m) * g.ldc + n;
        g.C[ci] = alpha * acc[i][j][r] + bsum + oldc[j][r];
      }
    }
  }
}

// One output tile of T x T (T = 128: 4 waves of 64 x 64; T = 64: 4 waves of 32 x 32), k-tiles of 64, two register stages of
// global prefetch.  128 x 128 when the output alone fills the chip; 64 x 64 for the long-K / small-output gradient GEMMs,
// where four times as many workgroups matter more than operand reuse.
template <int T, bool TN>
__device__ __forceinline__ void gemm_bf16_tile(const mmda_gemm_bf16_args& g, int splitk, int bx, int by, int sp, unsigned short* AB,
                                               float* slab, int ldn) {
  // two LDS buffers of (A block | B block): k-tile kt is computed out of buffer kt & 1 while tile kt + 1 is being stored into the
  // other one -- ONE workgroup barrier per k-tile, and the LDS stores (ds_write_b128 runs at a third of the read rate) sit beside
  // the other waves' MFMAs instead of between two barriers
  constexpr int LDK = TnLd<T>::v;             // tn: elements per k-row of an operand's LDS image
  constexpr int BUF = TN ? 2 * TK * LDK : 2 * T * LDT;
  unsigned short* As = AB;                   // (the epilogue stages the C tile over the first buffer)
  constexpr int W = T / 32;                  // MFMA tiles per wave per dimension
  constexpr int CH = T / 32;                 // 16-B chunks per thread per operand and k-tile (T rows x 8 chunks / 256 threads)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int row0 = by * T, col0 = bx * T;
  const int M = g.M, N = g.N, K = g.K;
  const int Kp = (K + 7) & ~7;
  const unsigned short* A = reinterpret_cast<const unsigned short*>(g.A);
  const unsigned short* Bm = reinterpret_cast<const unsigned short*>(g.B);
  const bool ones_row = g.bias_grad != nullptr;        // virtual all-ones row n == N of B: its output column is sum_k A[m,k]

  f32x4 acc[W][W];
#pragma unroll
  for (int i = 0; i < W; ++i)
#pragma unroll
    for (int j = 0; j < W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: T rows x 64 k = T*8 chunks of 8 bf16 per operand; chunk c: row = c >> 3, k = (c & 7) * 8.  NS register stages of global
  // prefetch: these problems leave one to four workgroups on a CU, so what hides the memory latency is loads in flight, not
  // occupancy -- two stages for the 128-tile (32 registers each), four for the 64-tile (16 each).
  constexpr int NS = T == 128 ? 2 : 4;
  u32x4 ra[NS][CH], rb[NS][CH];
  // Operand loads go through buffer descriptors with 32-bit byte offsets: rows past the matrix land beyond the descriptor's size and
  // read as zero in hardware, k past the (8-padded) depth is sent there by one select on the OFFSET -- no clamps, no 64-bit address
  // arithmetic and no selects on the loaded data in the k-loop (they were most of the wave's issue slots: the matrix pipe was 20 % busy
  // with the waves 38 % of their time in issue stalls, tools/prof_gemm_pmc.sh).  Host side guarantees (M + T) * lda * 2 < 4 GiB.
  // (tn: the descriptors end with the last k-row, so k >= K reads as zero; columns past the row's width read the next row -- finite
  //  values that only reach outputs nobody stores -- except past the very end, where they are zero again)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0,
      TN ? (int)((int64_t)K * g.lda * 2) : (int)(((int64_t)(M - 1) * g.lda + Kp) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Bm), 0,
      TN ? (int)((int64_t)K * g.ldb * 2) : (int)(((int64_t)(N - 1) * g.ldb + Kp) * 2), 0x00020000);
  constexpr unsigned OOB_OFF = 0xFFFFFF00u;
  constexpr int CPR = T / 8;                 // tn: 16-byte chunks per k-row of a tile
  unsigned offA[CH], offB[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = tid + 256 * i;
    if (TN) {
      const int kr = c / CPR, mc = (c % CPR) * 8;
      offA[i] = (unsigned)((kr * g.lda + row0 + mc) * 2);
      offB[i] = (unsigned)((kr * g.ldb + col0 + mc) * 2);
    } else {
      const int r = c >> 3, k = (c & 7) * 8;
      offA[i] = (unsigned)(((row0 + r) * g.lda + k) * 2);
      offB[i] = (unsigned)(((col0 + r) * g.ldb + k) * 2);
    }
  }
  auto load_tile = [&](u32x4 (&a)[CH], u32x4 (&b)[CH], int k0) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = tid + 256 * i;
      if (TN) {
        a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[i] + (unsigned)k0 * (unsigned)g.lda * 2u, 0, 0);
        b[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB[i] + (unsigned)k0 * (unsigned)g.ldb * 2u, 0, 0);
      } else {
        const bool k_ok = k0 + (c & 7) * 8 < Kp;
        a[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, k_ok ? offA[i] + (unsigned)k0 * 2u : OOB_OFF, 0, 0);
        b[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, k_ok ? offB[i] + (unsigned)k0 * 2u : OOB_OFF, 0, 0);
      }
    }
  };
  // the virtual all-ones row n == N of B (bias gradient) is written straight into the LDS tile that holds column N: it never
  // touches the global-load path (a lane-dependent branch there serialises the loads behind s_waitcnt vmcnt(0))
  const bool tile_has_ones = ones_row && col0 <= N && N < col0 + T;       // block-uniform
  auto store_tile = [&](const u32x4 (&a)[CH], const u32x4 (&b)[CH], int k0, int buf) {
    unsigned short* As = AB + buf * BUF;
    unsigned short* Bs = As + (TN ? TK * LDK : T * LDT);
    if (TN) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = tid + 256 * i;
        const int kr = c / CPR, mc = (c % CPR) * 8;
        *reinterpret_cast<u32x4*>(&As[kr * LDK + mc]) = a[i];
        u32x4 v = b[i];
        if (tile_has_ones && col0 + mc <= N && N < col0 + mc + 8) {      // the chunk that holds the virtual ones-column n == N
          const unsigned one = (k0 + kr < K) ? 0x3F80u : 0u;
          const int e = N - col0 - mc;
          unsigned w = v[e >> 1];
          w = (e & 1) ? ((w & 0x0000ffffu) | (one << 16)) : ((w & 0xffff0000u) | one);
          v[e >> 1] = w;
        }
        *reinterpret_cast<u32x4*>(&Bs[kr * LDK + mc]) = v;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = tid + 256 * i;
      const int r = c >> 3, k = (c & 7) * 8;
      *reinterpret_cast<u32x4*>(&As[r * LDT + k]) = a[i];        // (rows / k out of range arrived as zeros)
      u32x4 v = b[i];
      if (tile_has_ones) {                                     // block-uniform
        // bf16 1.0 = 0x3F80; elements past K stay zero so the sum runs over the real depth only
        unsigned e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = (k0 + k + q < K) ? 0x3F80u : 0u;
        const u32x4 ones = {e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
        v = (r == N - col0) ? ones : v;
      }
      *reinterpret_cast<u32x4*>(&Bs[r * LDT + k]) = v;
    }
  };

  const int nk_all = (K + TK - 1) / TK;
  const int per = (nk_all + splitk - 1) / splitk;
  const int kt0 = sp * per;
  const int nk = min(nk_all, kt0 + per);
  if (kt0 >= nk) return;
  const int fr = lane & 15, fq = lane >> 4;

#pragma unroll
  for (int u = 0; u < NS; ++u)
    if (kt0 + u < nk) load_tile(ra[u], rb[u], (kt0 + u) * TK);
  // tn: the fragment of a 16-column block for the 32-deep k-step ks.  ds_read_b64_tr_b16 takes, per 16-lane group, a block of 4 k-rows x
  // 16 columns -- lane 4q + p of the group gives the address of row q, columns 4p..4p+3 -- and hands lane i column i of the four rows.
  // Group fq reads k-rows 4 fq + q and 16 + 4 fq + q of the step: its eight k are not consecutive, but A and B fragments use the same
  // set, and a half-wave's two groups then touch eight CONSECUTIVE k-rows, which the row stride spreads over all 64 banks.
  auto tn_frag = [&](const unsigned short* Xs, int cbase, int ks) -> bf16x8 {
    typedef __attribute__((address_space(3))) v4s* lp;
    const int q = (lane & 15) >> 2, p = lane & 3;
    const unsigned short* a0 = &Xs[(ks * 32 + 4 * fq + q) * LDK + cbase + 4 * p];
    const v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)a0);
    const v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a0 + 16 * LDK));
    return bf16x8{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  };
  auto compute = [&](int buf) {
    const unsigned short* As = AB + buf * BUF;
    const unsigned short* Bs = As + (TN ? TK * LDK : T * LDT);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[W], b[W];
      if (TN) {
#pragma unroll
        for (int i = 0; i < W; ++i) {
          a[i] = tn_frag(As, wm * (T / 2) + i * 16, ks);
          b[i] = tn_frag(Bs, wn * (T / 2) + i * 16, ks);
        }
      } else
#pragma unroll
      for (int i = 0; i < W; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * (T / 2) + i * 16 + fr) * LDT + ks * 32 + fq * 8]);
        b[i] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * (T / 2) + i * 16 + fr) * LDT + ks * 32 + fq * 8]);
      }
#pragma unroll
      for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  // Unrolled by NS (even) so that register stages and LDS buffers are addressed statically: tile kt0 + u sits in stage u % NS and is
  // computed out of buffer u & 1.  Each round: store tile kt + 1 into the other buffer (last read for tile kt - 1, behind the
  // previous barrier), refill its stage with tile kt + 1 + NS, compute tile kt, barrier.
  store_tile(ra[0], rb[0], kt0 * TK, 0);
  if (kt0 + NS < nk) load_tile(ra[0], rb[0], (kt0 + NS) * TK);
  __syncthreads();
  for (int kt = kt0; kt < nk; kt += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if (kt + u < nk) {                                   // block-uniform
        const int nx = (u + 1) % NS;
        if (kt + u + 1 < nk) {
          store_tile(ra[nx], rb[nx], (kt + u + 1) * TK, (u + 1) & 1);
          if (kt + u + 1 + NS < nk) load_tile(ra[nx], rb[nx], (kt + u + 1 + NS) * TK);
        }
        compute(u & 1);
        __syncthreads();
      }
    }
  }

  gemm_bf16_epilogue<T>(g, acc, reinterpret_cast<float*>(As), row0, col0, splitk, sp, slab, ldn, tile_has_ones);
}

// One kernel per tile size: the 64 x 64 form needs half the registers and LDS of the 128 x 128 one, and the long-K gradient
// GEMMs that use it are bound by per-k-tile latency -- more resident workgroups per CU is what hides it.
// TN = false: nt problems only.  TN = true: either form per problem (a launch that holds tn problems runs every problem on this
// instance, so that input-gradient and weight-gradient GEMMs of a layer still go out together); held to the register budget of four
// workgroups per CU like the nt instance.
template <int T, bool TN>
__global__ __launch_bounds__(256, T == 64 ? 4 : 2) void gemm_bf16_kernel(Bf16Group G) {
  __shared__ __attribute__((aligned(16))) unsigned short AB[TN ? LdsElems<T>::tn : LdsElems<T>::nt];   // two buffers of (A block | B block); the C tile is staged over the first
  int pi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < G.n && (int)blockIdx.x >= G.start[k]) pi = k;
  const mmda_gemm_bf16_args& g = G.p[pi];
  const int splitk = G.splitk[pi];
  // XCD-aware tile order.  Workgroups are dealt round-robin over the eight XCDs (blocks b and b + 8 share one: observed, not
  // promised -- a wrong guess costs speed, not correctness), each with an L2 of its own.  Dealing tiles in block order put the tx
  // column tiles that share an A panel on tx DIFFERENT XCDs and every B panel on all eight: each L2 pulled every panel (176 MB
  // fetched per launch against 45 MB of operands, round 2's counters).  Here every problem starts at a multiple of eight blocks and
  // the blocks of one residue class walk ONE contiguous eighth of the problem's tile list (bx fastest, then by, then the K slice):
  // an A panel is read through one L2 only, a B panel through the L2s whose eighths reach it.
  const int tiles = G.tx[pi] * G.ty[pi] * splitk;
  const int local0 = (int)blockIdx.x - G.start[pi];
  const int x = local0 & 7, idx = local0 >> 3, q = tiles >> 3, r = tiles & 7;
  if (idx >= q + (x < r ? 1 : 0)) return;                                  // padding block of the last round of eight
  const int local = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  const int bx = local % G.tx[pi], by = (local / G.tx[pi]) % G.ty[pi], sp = local / (G.tx[pi] * G.ty[pi]);
  if (TN && g.tn) gemm_bf16_tile<T, true>(g, splitk, bx, by, sp, AB, G.slab[pi], G.ldn[pi]);      // block-uniform
  else gemm_bf16_tile<T, false>(g, splitk, bx, by, sp, AB, G.slab[pi], G.ldn[pi]);
}

// ------------------------------------------------------------------------------------------------ LDS-DMA pipelined 128 x 128 tile
// The large-batch form of the same GEMMs (round 2 measured the register-staged kernel above at 210 - 400 TFLOP/s on the B = 256
// problems with the matrix pipe 20 % busy: waves parked on memory and on issue).  Here the operand tiles go global -> LDS by LDS-DMA
// (`global_load_lds_dwordx4`: no staging registers, no ds_write instructions, no address arithmetic in the k-loop beyond one add per
// chunk), NS k-tiles deep, across ONE raw workgroup barrier per k-tile with a counted vmcnt (the DMA of tile kt + NS - 1 stays in
// flight while tile kt is multiplied):
//     wait until this wave's part of tile kt has landed (vmcnt) -> barrier (everybody's part has landed; everybody is done with
//     tile kt - 1) -> issue tile kt + NS - 1 into the stage tile kt - 1 was multiplied from -> multiply tile kt.
// An LDS-DMA writes wave-uniform base + lane x 16 bytes, so the LDS image is linear (no padding) and conflict-freedom comes from an
// XOR swizzle applied on BOTH sides: to the per-lane SOURCE address of the DMA and to the address of the fragment read.
//   nt image [128 rows][64 k] (128-byte rows): 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7).  A ds_read_b128 is served in
//      lane groups {0-3, 12-15, 20-27}, ...: eight rows at chunk c and eight at chunk c ^ 1 -- the swizzle sends them to sixteen
//      different 16-byte slots of the 256-byte bank line.
//   tn image [64 k][128 columns] (256-byte k-rows): chunk c of k-row k sits at chunk c ^ (2 (k & 7)).  A half-wave of the transposing
//      read (ds_read_b64_tr_b16, see tn_frag above) touches eight consecutive k-rows x 32 bytes: eight different 32-byte slots.
// Out-of-range pieces (rows past the matrix are clamped; k past the depth, k-rows past K) take their 16 bytes from a zero block in
// global memory: the select is on the source ADDRESS, nothing touches the data.  The bias gradient (column sums of A) is one extra
// MFMA per A fragment against an all-ones B fragment in the waves of the one column tile that holds column N.
__device__ __attribute__((aligned(16))) unsigned int g_zero_chunk[4] = {0u, 0u, 0u, 0u};
__device__ int g_dma_dbg = 0;      // tools only (mmda_debug_gemm_dma_mode): bit 0 = no MFMA work, bit 1 = no DMA after the prologue

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N == 0 || N == 6 || N == 8 || N == 12 || N == 16, "vmcnt immediates used by the DMA pipeline");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

// Epilogue of a TM x 128 tile held by TM / 32 waves (wm = w >> 1 row blocks of 64, wn = w & 1 column halves): the twin of
// gemm_bf16_epilogue for the DMA kernel's two tile heights (same staging through LDS, same slab / bias / gate-interleave handling).
template <int TM>
__device__ __forceinline__ void gemm_dma_epilogue(const mmda_gemm_bf16_args& g, f32x4 (&acc)[4][4], float* Cs, int row0, int col0,
                                                  int splitk, int sp, float* slab, int ldn, bool tile_has_ones) {
  constexpr int W = 4, TNn = 128, NTHR = TM * 2;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int M = g.M, N = g.N;
  const float alpha = g.alpha == 0.f ? 1.f : g.alpha;
  const bool to_slab = splitk > 1;                       // block-uniform
  if (!to_slab && tile_has_ones && wn == (N - col0) / 64) {
    // the column n == N holds sum_k A[m,k]: bias gradient(s).  One tile per row block holds that column: a single writer per entry.
    const int j = ((N - col0) % 64) / 16;
    if ((lane & 15) == (N - col0) % 16) {
#pragma unroll
      for (int i = 0; i < W; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
          float v = 0.f;
#pragma unroll
          for (int jj = 0; jj < W; ++jj) v = (jj == j) ? acc[i][jj][r] : v;
          if (m < M) {
            const int mo = g.perm_m_H > 0 ? gate_orig(m, g.perm_m_H) : m;
            g.bias_grad[mo] += v;
            if (g.bias_grad2) g.bias_grad2[mo] += v;
          }
        }
    }
  }
  const bool vec_out = to_slab || ((g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0 && (N & 3) == 0);      // block-uniform
  if (vec_out) {
    constexpr int LDC = TNn + 4;                         // floats per staged row
    constexpr int RP = 64;                               // rows per pass: one wave row block (33 KB of the ring)
    constexpr int C4 = TNn / 4;                          // float4 per staged row
    const int c4 = tid % C4;
    const int n = col0 + c4 * 4;
    const bool n_ok = n < (to_slab ? ldn : N);
    const float scale = to_slab ? 1.f : alpha;
    float* const out_base = to_slab ? slab + (int64_t)sp * M * ldn : g.C;
    const int out_ld = to_slab ? ldn : g.ldc;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    if (!to_slab && (g.bias || g.bias2)) {
      const int nn = min(n, N - 4);
      int nb0 = nn, nbs = 1;
      if (g.perm_n_H > 0) { const int G = 4 * g.perm_n_H, d = nn / G; nb0 = d * G + ((nn - d * G) >> 2); nbs = g.perm_n_H; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (g.bias) bsum[e] += g.bias[nb0 + e * nbs];
        if (g.bias2) bsum[e] += g.bias2[nb0 + e * nbs];
      }
    }
#pragma unroll
    for (int p = 0; p < TM / RP; ++p) {
      __syncthreads();                                   // operand tiles (first pass) / previous pass are no longer read
      if (wm == p) {                                     // wave-uniform: this pass is this wave's row block
#pragma unroll
        for (int i = 0; i < W; ++i)
#pragma unroll
          for (int j = 0; j < W; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Cs[(i * 16 + (lane >> 4) * 4 + r) * LDC + wn * 64 + j * 16 + (lane & 15)] = scale * acc[i][j][r];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RP * C4 / NTHR; ++q) {
        const int lr = (q * NTHR + tid) / C4;
        const int m = row0 + p * RP + lr;
        if (m < M && n_ok) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * LDC + c4 * 4]) + bsum;
          const int mo = (!to_slab && g.perm_m_H > 0) ? gate_orig(m, g.perm_m_H) : m;
          float* dst = out_base + (int64_t)mo * out_ld + n;
          if (!to_slab && g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = v;
        }
      }
    }
    return;
  }
  // Scalar path (unaligned rows)
#pragma unroll
  for (int i = 0; i < W; ++i) {
    float oldc[W][4];
    const bool rmw = g.accumulate;                             // block-uniform
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nc = min(col0 + wn * 64 + j * 16 + (lane & 15), N - 1);
        int mc = min(row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r, M - 1);
        if (g.perm_m_H > 0) mc = gate_orig(mc, g.perm_m_H);
        oldc[j][r] = rmw ? g.C[(int64_t)mc * g.ldc + nc] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int n = col0 + wn * 64 + j * 16 + (lane & 15);
      const bool n_ok = n < N;
      const int nc = min(n, N - 1);
      float bsum = 0.f;
      const int nb = g.perm_n_H > 0 ? gate_orig(nc, g.perm_n_H) : nc;
      if (g.bias) bsum += g.bias[nb];
      if (g.bias2) bsum += g.bias2[nb];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = row0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (!n_ok || m >= M) continue;
        const int64_t ci = (int64_t)(g.perm_m_H > 0 ? gate_orig(m, g.perm_m_H) : m) * g.ldc + n;
        g.C[ci] = alpha * acc[i][j][r] + bsum + oldc[j][r];
      }
    }
  }
}

// TM x 128 output tile, TM / 32 waves of 64 x 64 (TM = 128: 4 waves, 32 KB per stage, two workgroups per CU at NS = 2;
// TM = 256: 8 waves, 48 KB per stage, ONE workgroup per CU with NS = 3 -- 1.33x the FLOPs per byte moved into LDS and three k-tiles
// deep, for the long-K problems: what bounds the 128-row form is the DMA latency times the bytes the LDS can hold in flight).
template <bool TN, int NS, int TM>
__device__ __forceinline__ void gemm_bf16_dma_tile(const mmda_gemm_bf16_args& g, int splitk, int bx, int by, int sp, unsigned char* lds,
                                                   float* slab, int ldn) {
  constexpr int W = 4, NW = TM / 32;
  constexpr int OPA = TM * TK * 2, OPB = 128 * TK * 2;  // bytes of the operands' k-tile images
  constexpr int STAGE = OPA + OPB;
  constexpr int CA = 4, CB = 16 / NW;                  // DMA instructions per wave, operand and k-tile
  constexpr int ACPR = TM / 8;                         // tn: 16-byte chunks per k-row of the A image
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int row0 = by * TM, col0 = bx * 128;
  const int M = g.M, N = g.N, K = g.K;
  const int Kp = (K + 7) & ~7;
  const unsigned short* A = reinterpret_cast<const unsigned short*>(g.A);
  const unsigned short* Bm = reinterpret_cast<const unsigned short*>(g.B);
  const bool tile_has_ones = g.bias_grad != nullptr && col0 <= N && N < col0 + 128;     // block-uniform
  const bool my_ones = tile_has_ones && wn == (N - col0) / 64;                          // wave-uniform

  f32x4 acc[W][W], accb[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- source addressing of this thread's CA + CB chunks per k-tile.  DMA instruction j of wave w fills LDS bytes
  // [(C w + j) * 1024, +1024) of an operand image: lane l writes slot s = (C w + j) * 64 + l.  Element offsets are 32-bit (the host
  // checks the operand sizes).
  int offA[CA], offB[CB];
  int kofA[CA], kofB[CB];                              // nt: first k of the chunk within the tile; tn: k-row within the tile
#pragma unroll
  for (int j = 0; j < CA; ++j) {
    const int s = (CA * w + j) * 64 + lane;
    if (TN) {
      const int kr = s / ACPR, c = (s % ACPR) ^ (2 * (kr & 7));
      kofA[j] = kr;
      offA[j] = kr * g.lda + row0 + c * 8;
    } else {
      const int r = s >> 3, c = (s & 7) ^ ((r >> 1) & 7);
      kofA[j] = c * 8;
      offA[j] = min(row0 + r, M - 1) * g.lda + c * 8;
    }
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int s = (CB * w + j) * 64 + lane;
    if (TN) {
      const int kr = s >> 4, c = (s & 15) ^ (2 * (kr & 7));
      kofB[j] = kr;
      offB[j] = kr * g.ldb + col0 + c * 8;
    } else {
      const int r = s >> 3, c = (s & 7) ^ ((r >> 1) & 7);
      kofB[j] = c * 8;
      offB[j] = min(col0 + r, N - 1) * g.ldb + c * 8;
    }
  }
  // tn: the last element offset a 16-byte read may start at (columns past a row's width read on into the next row -- finite values
  // that only reach outputs nobody stores -- but nothing may read past the end of the operand)
  const int limA = TN ? K * g.lda - 8 : 0, limB = TN ? K * g.ldb - 8 : 0;
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero_chunk);

  const int nk_all = (K + TK - 1) / TK;
  const int per = (nk_all + splitk - 1) / splitk;
  const int kt0 = sp * per;
  const int nk = min(nk_all, kt0 + per);
  if (kt0 >= nk) return;                               // (the host leaves no empty slice)

  // The DMA goes out as inline assembly: a `__builtin_amdgcn_global_load_lds` the compiler knows about makes it put
  // `s_waitcnt vmcnt(0)` in front of the next LDS read (it cannot tell the stages of the ring apart), which is the end of the
  // pipelining; what orders an LDS read behind a DMA here is the counted vmcnt + barrier of the k-loop, by construction.
  // m0 = LDS base of the instruction (wave-uniform); the lane's 16 bytes land at m0 + lane * 16.
  auto dma16 = [&](const unsigned short* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue = [&](int kt, int stage) {                // kt >= nk: a tile of zeros nobody reads (keeps the vmcnt arithmetic uniform)
    const int k0 = kt * TK;
    const bool live = kt < nk;
    const unsigned dA = lds0 + stage * STAGE + (CA * w) * 1024;
    const unsigned dB = lds0 + stage * STAGE + OPA + (CB * w) * 1024;
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      int oa;
      bool oka;
      if (TN) { oa = offA[j] + k0 * g.lda; oka = live & (k0 + kofA[j] < K) & (oa <= limA); }
      else { oa = offA[j] + k0; oka = live & (k0 + kofA[j] < Kp); }
      const unsigned short* pa = A + oa;
      pa = oka ? pa : zero;
      dma16(pa, dA + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      int ob;
      bool okb;
      if (TN) { ob = offB[j] + k0 * g.ldb; okb = live & (k0 + kofB[j] < K) & (ob <= limB); }
      else { ob = offB[j] + k0; okb = live & (k0 + kofB[j] < Kp); }
      const unsigned short* pb = Bm + ob;
      pb = okb ? pb : zero;
      dma16(pb, dB + j * 1024);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  auto compute = [&](int stage) {
    const unsigned char* As = lds + stage * STAGE;
    const unsigned char* Bs = As + OPA;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[W], b[W];
      if (TN) {
        typedef __attribute__((address_space(3))) v4s* lp;
        const int k = ks * 32 + 4 * fq + tq;
        const int sw = 2 * (k & 7);
#pragma unroll
        for (int i = 0; i < W; ++i) {
          const int ca = (wm * 64 + i * 16) >> 3, cb = (wn * 64 + i * 16) >> 3;
          const unsigned char* pa = As + k * (2 * TM) + (((ca + (tp >> 1)) ^ sw) << 4) + 8 * (tp & 1);
          const unsigned char* pb = Bs + k * 256 + (((cb + (tp >> 1)) ^ sw) << 4) + 8 * (tp & 1);
          const v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)pa), a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(pa + 16 * 2 * TM));
          const v4s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)pb), b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(pb + 16 * 256));
          a[i] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          b[i] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        }
      } else {
        const int co = ((ks * 4 + fq) ^ (fr >> 1)) << 4;
#pragma unroll
        for (int i = 0; i < W; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + i * 16 + fr) * 128 + co);
          b[i] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + i * 16 + fr) * 128 + co);
        }
      }
#pragma unroll
      for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (my_ones) {
#pragma unroll
        for (int i = 0; i < W; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
      }
    }
  };

  const int dbg = g_dma_dbg;
#pragma unroll
  for (int u = 0; u < NS - 1; ++u) issue(kt0 + u, u);
  for (int kt = kt0; kt < nk; kt += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if (kt + u < nk) {                                 // block-uniform
        wait_vmcnt<(CA + CB) * (NS - 2)>();              // this wave's part of tile kt + u has landed
        __builtin_amdgcn_s_barrier();                    // ... everybody's has; and everybody is done reading tile kt + u - 1
        asm volatile("" ::: "memory");
        if (!(dbg & 2)) issue(kt + u + NS - 1, (u + NS - 1) % NS);
        if (!(dbg & 1)) compute(u);
      }
    }
  }
  wait_vmcnt<0>();                                       // the trailing zero tiles must have landed before the C tile is staged over them
  if (my_ones) {
    const int jo = ((N - col0) % 64) / 16;
    if ((lane & 15) == (N - col0) % 16) {
#pragma unroll
      for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (j == jo) acc[i][j] = accb[i];
    }
  }
  if (TM == 128) gemm_bf16_epilogue<128>(g, acc, reinterpret_cast<float*>(lds), row0, col0, splitk, sp, slab, ldn, tile_has_ones);
  else gemm_dma_epilogue<TM>(g, acc, reinterpret_cast<float*>(lds), row0, col0, splitk, sp, slab, ldn, tile_has_ones);
}

template <int NS, int TM>
__global__ __launch_bounds__(TM * 2, (TM == 128 && NS == 2) ? 2 : 1) void gemm_bf16_dma_kernel(Bf16Group G) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * (TM + 128) * TK * 2];      // the ONLY LDS object of the kernel
  int pi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < G.n && (int)blockIdx.x >= G.start[k]) pi = k;
  const mmda_gemm_bf16_args& g = G.p[pi];
  const int splitk = G.splitk[pi];
  const int tiles = G.tx[pi] * G.ty[pi] * splitk;          // XCD-aware order: see gemm_bf16_kernel
  const int local0 = (int)blockIdx.x - G.start[pi];
  const int x = local0 & 7, idx = local0 >> 3, q = tiles >> 3, r = tiles & 7;
  if (idx >= q + (x < r ? 1 : 0)) return;
  const int local = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  const int bx = local % G.tx[pi], by = (local / G.tx[pi]) % G.ty[pi], sp = local / (G.tx[pi] * G.ty[pi]);
  if (g.tn) gemm_bf16_dma_tile<true, NS, TM>(g, splitk, bx, by, sp, lds, G.slab[pi], G.ldn[pi]);      // block-uniform
  else gemm_bf16_dma_tile<false, NS, TM>(g, splitk, bx, by, sp, lds, G.slab[pi], G.ldn[pi]);
}

__global__ __launch_bounds__(256) void convert_kernel(ConvLaunch L) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][66];
  convert_block(L, (int)blockIdx.x, tile);
}

}  // namespace

extern "C" int mmda_gemm_bf16_grouped(const mmda_gemm_bf16_args* args, int n, void* stream) {
  if (!args || n < 0) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    const mmda_gemm_bf16_args& a = args[i];
    if (!a.A || !a.B || !a.C || a.M < 0 || a.N < 0 || a.K <= 0) return MMDA_EINVAL;
    if (a.tn) {
      if ((a.lda & 3) || (a.ldb & 3) || (((uintptr_t)a.A | (uintptr_t)a.B) & 3) || a.lda < a.M || a.ldb < a.N) return MMDA_EINVAL;
      if (((double)a.K + 64.0) * a.lda * 2.0 >= 4.0e9 || ((double)a.K + 64.0) * a.ldb * 2.0 >= 4.0e9) return MMDA_EINVAL;
      if (a.perm_n_H < 0 || a.perm_m_H < 0 || (a.perm_n_H && a.N % (4 * a.perm_n_H)) || (a.perm_m_H && a.M % (4 * a.perm_m_H))) return MMDA_EINVAL;
      continue;
    }
    if ((a.lda & 7) || (a.ldb & 7) || (((uintptr_t)a.A | (uintptr_t)a.B) & 15)) return MMDA_EINVAL;
    if (a.lda < ((a.K + 7) & ~7) || a.ldb < ((a.K + 7) & ~7)) return MMDA_EINVAL;
    // operands are addressed with 32-bit byte offsets through buffer descriptors (row clamping is done by the descriptor's size)
    if (((double)a.M + 128.0) * a.lda * 2.0 >= 4.0e9 || ((double)a.N + 129.0) * a.ldb * 2.0 >= 4.0e9) return MMDA_EINVAL;
    if (a.perm_n_H < 0 || a.perm_m_H < 0 || (a.perm_n_H && a.N % (4 * a.perm_n_H)) || (a.perm_m_H && a.M % (4 * a.perm_m_H))) return MMDA_EINVAL;
  }
  // Three kernel classes.  0: 64 x 64 register-staged (small / unaligned problems), 1: 128 x 128 register-staged, 2: 128 x 128
  // LDS-DMA pipelined (gemm_bf16_dma_kernel).  Class 2 takes every problem whose operands can be moved by 16-byte LDS-DMA (nt: always
  // -- rows are 16-byte aligned by contract; tn: leading dimensions multiples of 8 and 16-byte aligned bases) and whose output is at
  // least MMDA_GEMM_DMA_MIN (default 96) rows and columns; MMDA_GEMM_DMA=0 switches it off, MMDA_GEMM_DMA_STAGES=2|3 sets the depth of
  // its LDS ring (2: 64 KB, two workgroups per CU; 3: 96 KB, one).
  static const int t128_min = getenv("MMDA_GEMM_T128_MIN") ? atoi(getenv("MMDA_GEMM_T128_MIN")) : 512;     // experiment switch
  static const int dma_on = getenv("MMDA_GEMM_DMA") ? atoi(getenv("MMDA_GEMM_DMA")) : 1;
  static const int dma_min = getenv("MMDA_GEMM_DMA_MIN") ? atoi(getenv("MMDA_GEMM_DMA_MIN")) : 96;
  static const int dma_stages = getenv("MMDA_GEMM_DMA_STAGES") ? atoi(getenv("MMDA_GEMM_DMA_STAGES")) : 2;
  // ... and only in a call of large-batch problems -- some problem with >= 8192 rows (nt) or k-rows (tn): T * B of the step.  Measured
  // (step, ms; DMA class on / off): B=32 0.694 / 0.657, B=64 0.834 / 0.822, B=128 1.167 / 1.174, B=256 1.85 / 2.02 -- below that the
  // problems are a few k-tiles on a few hundred workgroups, where the register-staged 64 x 64 kernel at four workgroups per CU is
  // ahead.  MMDA_GEMM_DMA_MIN_ROWS moves the limit.
  static const int dma_min_rows = getenv("MMDA_GEMM_DMA_MIN_ROWS") ? atoi(getenv("MMDA_GEMM_DMA_MIN_ROWS")) : 8192;
  int call_rows = 0;
  for (int i = 0; i < n; ++i) call_rows = max(call_rows, args[i].tn ? args[i].K : args[i].M);
  // (experiment switch MMDA_GEMM_DMA_FWD=1: calls of forward products only -- nothing accumulates -- take the DMA class at any size)
  static const int dma_fwd = getenv("MMDA_GEMM_DMA_FWD") ? atoi(getenv("MMDA_GEMM_DMA_FWD")) : 0;
  bool fwd_only = n > 0;
  for (int i = 0; i < n; ++i) fwd_only = fwd_only && !args[i].accumulate && !args[i].tn && !args[i].bias_grad;
  const bool dma_call = dma_on && (call_rows >= dma_min_rows || (dma_fwd && fwd_only));
  // (OFF by default since the end of round 3: with the rest of the step as it is now the B=256 step measures 1.690 ms without the
  //  class against 1.705 with it -- its 147 KB workgroups do not fit a CU beside a recurrent kernel's, and on the main stream they
  //  are no faster than two 128-row workgroups per CU; MMDA_GEMM_DMA_TALL=1 switches it on)
  static const int dma_tall = getenv("MMDA_GEMM_DMA_TALL") ? atoi(getenv("MMDA_GEMM_DMA_TALL")) : 0;
  static const int tall_stages = getenv("MMDA_GEMM_DMA_TALL_STAGES") ? atoi(getenv("MMDA_GEMM_DMA_TALL_STAGES")) : 3;
  auto class_of = [&](const mmda_gemm_bf16_args& a) {
    const int Ne = a.N + (a.bias_grad ? 1 : 0);
    if (dma_call && a.M >= dma_min && Ne >= dma_min) {
      bool ok = true;
      if (a.tn) ok = !(a.lda & 7) && !(a.ldb & 7) && !(((uintptr_t)a.A | (uintptr_t)a.B) & 15) && ((double)a.K + 64.0) * (double)max(a.lda, a.ldb) < 2.0e9;
      // 256-row tiles (class 3) for the long k-walks of a tall output: K >= 1024, M >= 512 (input gradients: K = 8H; weight gradients:
      // K = T * B).  The short-K forward products stay on the 128-row form: they are a prologue and an epilogue around five to ten
      // k-tiles, and two workgroups per CU overlap those where one cannot.  MMDA_GEMM_DMA_TALL=0 switches the class off.
      if (ok && dma_tall && a.K >= 1024 && a.M >= 512) return 3;
      if (ok) return 2;
    }
    return ceil_div(Ne, 128) * ceil_div(a.M, 128) >= t128_min ? 1 : 0;
  };
  // ---- plan: per class, the problems in launch order and their split-K
  struct Plan { std::vector<int> order; std::vector<int> sks; };
  constexpr int NCLASS = 4;
  Plan plan[NCLASS];
  // (T: the class id stands for its tile -- 64 x 64, 128 x 128, 128 x 128, 256 x 128 rows x columns)
  auto tiles_of = [&](const mmda_gemm_bf16_args& a, int ci) {
    const int tm = ci == 0 ? 64 : (ci == 3 ? 256 : 128), tn = ci == 0 ? 64 : 128;
    return ceil_div(a.N + (a.bias_grad ? 1 : 0), tn) * ceil_div(a.M, tm);
  };
  for (int ci = 0; ci < NCLASS; ++ci) {
    const int T = ci;
    std::vector<int>& order = plan[ci].order;
    std::vector<int>& sks = plan[ci].sks;
    sks.assign(n, 1);
    // Workgroups are dealt in block order: the problems with the longest K loops go first, so that their workgroups do not
    // form the tail of the launch.  `crowded`: the launch fills the chip twice over without any split-K.
    int64_t all_tiles = 0;
    for (int i = 0; i < n; ++i) {
      const mmda_gemm_bf16_args& a = args[i];
      if (a.M == 0 || a.N == 0 || class_of(a) != ci) continue;
      order.push_back(i);
      all_tiles += tiles_of(a, T);
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return args[x].K > args[y].K; });
    // workgroups of this class the chip holds at once
    const int resident = ci == 0 ? 1024 : (ci == 1 ? 512 : (ci == 3 ? 256 : (dma_stages == 2 ? 512 : 256)));
    const bool crowded = all_tiles >= (ci >= 2 ? resident : 512);
    int64_t work2 = 0;                                    // class 2: k-tiles of the whole launch
    for (int i : order) work2 += (int64_t)tiles_of(args[i], T) * ceil_div(args[i].K, TK);
    // pass 1: the split of every problem on its own
    static const int split_min_nk = getenv("MMDA_GEMM_SPLIT_MIN_NK") ? atoi(getenv("MMDA_GEMM_SPLIT_MIN_NK")) : 128;
    for (int i : order) {
      const mmda_gemm_bf16_args& a = args[i];
      const int tiles = tiles_of(a, T);
      const int nk = ceil_div(a.K, TK);
      const double out_mb = (double)a.M * a.N * 4.0 / 1048576.0;
      int sk = 1;
      if (ci >= 2) {
        // the DMA kernel: a workgroup's time is its k-tiles (~1 us each: the latency of the tile in flight), so the launch is cut
        // into pieces of equal length -- target = the k-tiles per workgroup at which the whole launch fills the resident slots once --
        // and a problem is split where its k-walk is longer than that (the weight gradients of a large batch, K = T * B, beside
        // the input gradients, K = 8H, of the same launch).  Every slice keeps >= 6 k-tiles (prologue, the C tile's trip through
        // the slab), slabs stay <= 32 MB per problem.
        const int target = (int)std::max<int64_t>(8, ceil_div64(work2, resident));
        sk = (nk + target / 2) / target;
        if (sk > nk / 6) sk = nk / 6;
        while (sk > 1 && sk * out_mb > 32.0) --sk;
        if (sk > 32) sk = 32;
        if (sk < 1) sk = 1;
      } else if (tiles < 1024 && nk >= 128) {
        // Register-staged classes: split only a very long k-walk (K >= 8192: the small-modality weight gradients of a large batch, a
        // handful of tiles walking T * B rows) on fewer workgroups than the chip holds; every k-tile is one memory latency, so the walk
        // is cut until the slots are full.  Below that nothing is split any more: with the slices combined through slabs and a reduce
        // launch (round 3) instead of float atomics, a split costs the MOSEI-sized problems more than its parallelism returns --
        // measured, alternating runs (step, ms; split / no split): B=16 0.621 / 0.612, B=32 0.665 / 0.652, B=64 0.846 / 0.842,
        // B=128 1.206 / 1.196.  MMDA_GEMM_SPLIT_MIN_NK moves the limit (8: round 2's policy).
        static const int long_k = getenv("MMDA_GEMM_LONGK_SPLIT") ? atoi(getenv("MMDA_GEMM_LONGK_SPLIT")) : 1;
        sk = long_k ? ceil_div(1024, tiles) : 1;
        while (sk > 1 && (nk / sk < 16 || sk * out_mb > 24.0)) --sk;
      } else if (split_min_nk < 128 && tiles < 256 && nk >= split_min_nk) {
        sk = ceil_div(512, tiles);
        if (sk > nk / 4) sk = nk / 4;
        while (sk > 1 && sk * out_mb > 6.0) --sk;
        if (sk > 16) sk = 16;
        if (sk < 1) sk = 1;
      }
      static const int max_split = getenv("MMDA_GEMM_MAX_SPLIT") ? atoi(getenv("MMDA_GEMM_MAX_SPLIT")) : 0;      // experiment switch
      if (max_split > 0 && sk > max_split) sk = max_split;
      // a fresh (non-accumulated) output gains from a split only with a long K loop in a launch that would otherwise leave the chip
      // underfilled
      if (sk > 1 && !a.accumulate && (crowded || nk < 16)) sk = 1;
      sks[i] = sk;
    }
    // pass 2: the launch as a whole.  The chip holds SLOTS workgroups of this kernel at once; a launch of 1.x times that runs a second,
    // mostly empty round.  While the launch sits between one and two rounds, the most finely split problems give slices back.
    {
      static const int slots = getenv("MMDA_GEMM_SLOTS") ? atoi(getenv("MMDA_GEMM_SLOTS")) : -1;
      const int sl = slots >= 0 ? slots : resident;
      auto total = [&]() { int64_t t = 0; for (int i : order) t += (int64_t)tiles_of(args[i], T) * sks[i]; return t; };
      int64_t tot = total();
      while (ci < 2 && sl > 0 && tot > sl && tot < 2 * (int64_t)sl) {
        int best = -1;
        for (int i : order) if (sks[i] > 1 && (best < 0 || sks[i] > sks[best] || (sks[i] == sks[best] && tiles_of(args[i], T) > tiles_of(args[best], T)))) best = i;
        if (best < 0) break;
        --sks[best];
        tot = total();
      }
    }
    // no empty slices (an empty slice would leave its slab unwritten): sk = the number of slices that hold k-tiles
    for (int i : order) {
      const int nk = ceil_div(args[i].K, TK);
      const int per = ceil_div(nk, sks[i]);
      sks[i] = ceil_div(nk, per);
    }
  }
  // ---- slabs of the split problems: one scratch request for the whole call
  std::vector<int64_t> slab_off(n, -1);
  int64_t slab_floats = 0;
  for (int ci = 0; ci < NCLASS; ++ci)
    for (int i : plan[ci].order)
      if (plan[ci].sks[i] > 1) {
        const mmda_gemm_bf16_args& a = args[i];
        const int ldn = round_up(a.N + (a.bias_grad ? 1 : 0), 4);
        slab_off[i] = slab_floats;
        slab_floats += (int64_t)plan[ci].sks[i] * a.M * ldn;
      }
  float* slab_base = nullptr;
  if (slab_floats > 0) {
    slab_base = mmda_scratch_get(s, (size_t)slab_floats * sizeof(float));
    if (!slab_base) return MMDA_ELAUNCH;
  }
  // ---- launches (the DMA class first: it holds the largest problems), one reduce launch behind them all
  std::vector<SplitKJob> jobs;
  for (int cc = 0; cc < NCLASS; ++cc) {
    const int ci = cc == 0 ? 3 : (cc == 1 ? 2 : cc - 2);
    const int T = ci == 0 ? 64 : 128;                      // tile columns; rows: 256 for class 3
    const int TMr = ci == 3 ? 256 : T;
    const std::vector<int>& order = plan[ci].order;
    const std::vector<int>& sks = plan[ci].sks;
    bool any_tn = false;
    for (int i : order) any_tn = any_tn || args[i].tn;
    const int form = any_tn ? 1 : 0;
    Bf16Group G;
    G.n = 0;
    int blocks = 0;
    auto flush = [&]() -> int {
      if (blocks == 0) { G.n = 0; return MMDA_OK; }
      for (int k = G.n; k <= GROUP_MAX; ++k) G.start[k] = blocks;
      for (int k = G.n; k < GROUP_MAX; ++k) { G.p[k] = G.p[0]; G.tx[k] = G.ty[k] = G.splitk[k] = 1; G.tile[k] = T; G.slab[k] = nullptr; G.ldn[k] = 0; }
      if (ci == 3) {
        if (tall_stages == 2) hipLaunchKernelGGL((gemm_bf16_dma_kernel<2, 256>), dim3(blocks), dim3(512), 0, s, G);
        else hipLaunchKernelGGL((gemm_bf16_dma_kernel<3, 256>), dim3(blocks), dim3(512), 0, s, G);
      } else if (ci == 2) {
        if (dma_stages == 3) hipLaunchKernelGGL((gemm_bf16_dma_kernel<3, 128>), dim3(blocks), dim3(256), 0, s, G);
        else hipLaunchKernelGGL((gemm_bf16_dma_kernel<2, 128>), dim3(blocks), dim3(256), 0, s, G);
      } else if (form == 0) {
        if (T == 128) hipLaunchKernelGGL((gemm_bf16_kernel<128, false>), dim3(blocks), dim3(256), 0, s, G);
        else hipLaunchKernelGGL((gemm_bf16_kernel<64, false>), dim3(blocks), dim3(256), 0, s, G);
      } else {
        if (T == 128) hipLaunchKernelGGL((gemm_bf16_kernel<128, true>), dim3(blocks), dim3(256), 0, s, G);
        else hipLaunchKernelGGL((gemm_bf16_kernel<64, true>), dim3(blocks), dim3(256), 0, s, G);
      }
      MMDA_CHECK_LAUNCH("mmda_gemm_bf16_grouped");
      G.n = 0; blocks = 0;
      return MMDA_OK;
    };
    for (int i : order) {
      const mmda_gemm_bf16_args& a = args[i];
      if (G.n == GROUP_MAX) { int rc = flush(); if (rc) return rc; }
      const int k = G.n++;
      G.p[k] = a;
      const int Ne = a.N + (a.bias_grad ? 1 : 0);
      G.tile[k] = T;
      G.tx[k] = ceil_div(Ne, T); G.ty[k] = ceil_div(a.M, TMr);
      const int tiles = G.tx[k] * G.ty[k];
      const int sk = sks[i];
      G.splitk[k] = sk;
      G.slab[k] = nullptr; G.ldn[k] = 0;
      if (sk > 1) {
        G.ldn[k] = round_up(Ne, 4);
        G.slab[k] = slab_base + slab_off[i];
        SplitKJob J = {};
        J.slab = G.slab[k]; J.C = a.C; J.M = a.M; J.N = a.N; J.ldn = G.ldn[k]; J.ldc = a.ldc; J.sk = sk; J.batch = 1;
        J.alpha = a.alpha; J.bias = a.bias; J.bias2 = a.bias2; J.bias_grad = a.bias_grad; J.bias_grad2 = a.bias_grad2;
        J.accumulate = a.accumulate; J.perm_m_H = a.perm_m_H; J.perm_n_H = a.perm_n_H;
        jobs.push_back(J);
      }
      G.start[k] = blocks;
      blocks += round_up(tiles * sk, 8);                 // every problem starts at a multiple of eight blocks (XCD-aware order)
    }
    int rc = flush();
    if (rc) return rc;
  }
  if (!jobs.empty()) { const int rc = mmda_splitk_reduce(jobs.data(), (int)jobs.size(), s); if (rc) return rc; }
  return MMDA_OK;
}

extern "C" int mmda_debug_gemm_dma_mode(int mode) {       // tools/ only: ablate the DMA kernel's k-loop (results are then wrong)
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dma_dbg), &mode, sizeof(int)) == hipSuccess ? MMDA_OK : MMDA_ELAUNCH;
}

extern "C" int mmda_convert_bf16(const mmda_convert_job* jobs, int n, void* stream) {
  if (!jobs || n < 0) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n; base += CONV_MAX) {
    const int cnt = (n - base) < CONV_MAX ? (n - base) : CONV_MAX;
    ConvLaunch L;
    int blocks = 0;
    const int rc = conv_build(jobs + base, cnt, L, blocks);
    if (rc) return rc;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(convert_kernel, dim3(blocks), dim3(256), 0, s, L);
    MMDA_CHECK_LAUNCH("mmda_convert_bf16");
  }
  return MMDA_OK;
}
