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
  int start[GROUP_MAX + 1];
  int tx[GROUP_MAX], ty[GROUP_MAX], splitk[GROUP_MAX], tile[GROUP_MAX];
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

// One output tile of T x T (T = 128: 4 waves of 64 x 64; T = 64: 4 waves of 32 x 32), k-tiles of 64, two register stages of
// global prefetch.  128 x 128 when the output alone fills the chip; 64 x 64 for the long-K / small-output gradient GEMMs,
// where four times as many workgroups matter more than operand reuse.
template <int T, bool TN>
__device__ __forceinline__ void gemm_bf16_tile(const mmda_gemm_bf16_args& g, int splitk, int bx, int by, int sp, unsigned short* AB) {
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

  const float alpha = g.alpha == 0.f ? 1.f : g.alpha;
  if (tile_has_ones && wn == (N - col0) / (T / 2)) {
    // the column n == N holds sum_k A[m,k]: bias gradient(s)
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
            atomicAdd(&g.bias_grad[mo], v);
            if (g.bias_grad2) atomicAdd(&g.bias_grad2[mo], v);
          }
        }
    }
  }
  // ---- C tile.  Without split-K and with 16-byte-aligned rows the tile goes out through LDS: the accumulator fragments hold 4
  // rows x 16 columns per wave-register, so storing them directly is 64-byte pieces (a 1600 x 2400 fp32 output took 16 us
  // that way, 4x a memset of the same size); staged, every store instruction writes 16 bytes per lane along a row and the
  // bias / accumulate reads are 16-byte loads of the same shape.
  const bool vec_out = splitk == 1 && (g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0 && (N & 3) == 0;      // block-uniform
  if (vec_out) {
    constexpr int LDC = T + 4;                           // floats per staged row
    constexpr int RP = T == 128 ? 32 : 64;               // rows per pass: RP * LDC * 4 bytes <= the operand LDS block
    constexpr int C4 = T / 4;                            // float4 per staged row; divides 256, so a thread keeps ONE column group
    float* Cs = reinterpret_cast<float*>(As);
    const int c4 = tid % C4;
    const int n = col0 + c4 * 4;
    const bool n_ok = n < N;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    if (g.bias || g.bias2) {
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
              Cs[(lr0 + (lane >> 4) * 4 + r) * LDC + wn * (T / 2) + j * 16 + (lane & 15)] = alpha * acc[i][j][r];
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RP * C4 / 256; ++q) {
        const int lr = (q * 256 + tid) / C4;
        const int m = row0 + p * RP + lr;
        if (m < M && n_ok) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[lr * LDC + c4 * 4]) + bsum;
          const int mo = g.perm_m_H > 0 ? gate_orig(m, g.perm_m_H) : m;
          float* dst = g.C + (int64_t)mo * g.ldc + n;
          if (g.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = v;
        }
      }
    }
    return;
  }
  // Scalar path (split-K atomics, unaligned rows).  Accumulating without split-K reads C first: all loads are issued from
  // clamped addresses before the first add (a load under the m < M / n < N branches would be waited for one by one).
#pragma unroll
  for (int i = 0; i < W; ++i) {
    float oldc[W][4];
    const bool rmw = g.accumulate && splitk == 1;              // block-uniform
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
        if (splitk > 1) atomicAdd(&g.C[ci], alpha * acc[i][j][r] + (sp == 0 ? bsum : 0.f));
        else g.C[ci] = alpha * acc[i][j][r] + bsum + oldc[j][r];
      }
    }
  }
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
  const int local = blockIdx.x - G.start[pi];
  const int bx = local % G.tx[pi], by = (local / G.tx[pi]) % G.ty[pi], sp = local / (G.tx[pi] * G.ty[pi]);
  if (TN && g.tn) gemm_bf16_tile<T, true>(g, splitk, bx, by, sp, AB);      // block-uniform
  else gemm_bf16_tile<T, false>(g, splitk, bx, by, sp, AB);
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
  // 128 x 128 tiles when they alone fill the chip twice over; else 64 x 64 (4x the workgroups, 2x the residency)
  static const int t128_min = getenv("MMDA_GEMM_T128_MIN") ? atoi(getenv("MMDA_GEMM_T128_MIN")) : 512;     // experiment switch
  auto tile_of = [](const mmda_gemm_bf16_args& a) {
    const int Ne = a.N + (a.bias_grad ? 1 : 0);
    return ceil_div(Ne, 128) * ceil_div(a.M, 128) >= t128_min ? 128 : 64;
  };
  for (int T = 64; T <= 128; T += 64) {
    bool any_tn = false;
    for (int i = 0; i < n; ++i) any_tn = any_tn || (args[i].tn && args[i].M > 0 && args[i].N > 0 && tile_of(args[i]) == T);
    const int form = any_tn ? 1 : 0;
    Bf16Group G;
    G.n = 0;
    int blocks = 0;
    auto flush = [&]() -> int {
      if (blocks == 0) { G.n = 0; return MMDA_OK; }
      for (int k = G.n; k <= GROUP_MAX; ++k) G.start[k] = blocks;
      for (int k = G.n; k < GROUP_MAX; ++k) { G.p[k] = G.p[0]; G.tx[k] = G.ty[k] = G.splitk[k] = 1; G.tile[k] = T; }
      if (form == 0) {
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
    // Workgroups are dealt in block order: the problems with the longest K loops go first, so that their workgroups do not
    // form the tail of the launch.  `crowded`: the launch fills the chip twice over without any split-K.
    std::vector<int> order;
    int64_t all_tiles = 0;
    for (int i = 0; i < n; ++i) {
      const mmda_gemm_bf16_args& a = args[i];
      if (a.M == 0 || a.N == 0 || tile_of(a) != T) continue;
      order.push_back(i);
      all_tiles += (int64_t)ceil_div(a.N + (a.bias_grad ? 1 : 0), T) * ceil_div(a.M, T);
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return args[x].K > args[y].K; });
    const bool crowded = all_tiles >= 512;
    // pass 1: the split of every problem on its own
    std::vector<int> sks(n, 1);
    auto tiles_of = [&](const mmda_gemm_bf16_args& a) { return ceil_div(a.N + (a.bias_grad ? 1 : 0), T) * ceil_div(a.M, T); };
    for (int i : order) {
      const mmda_gemm_bf16_args& a = args[i];
      const int tiles = tiles_of(a);
      const int nk = ceil_div(a.K, TK);
      // split-K combines through float atomics, which the chip retires at ~1.3 TB/s of added bytes: split only while the
      // added bytes stay small (<= 6 MB, ~5 us) and every slice keeps >= 4 k-tiles
      int sk = 1;
      if (tiles < 256 && nk >= 8) {
        sk = ceil_div(512, tiles);
        if (sk > nk / 4) sk = nk / 4;
        const double out_mb = (double)a.M * a.N * 4.0 / 1048576.0;
        while (sk > 1 && sk * out_mb > 6.0) --sk;
        if (sk > 16) sk = 16;
        if (sk < 1) sk = 1;
      } else if (tiles < 1024 && nk >= 64) {
        // a long k-walk (the weight-gradient GEMMs of a large batch: K = T * B) on fewer workgroups than the chip holds (four per CU):
        // every k-tile is one memory latency, so the walk is cut until the slots are full -- while the atomic combine stays a small
        // part of the walk it shortens (added bytes at ~1.3 TB/s against ~0.7 us per k-tile saved)
        static const int long_k = getenv("MMDA_GEMM_LONGK_SPLIT") ? atoi(getenv("MMDA_GEMM_LONGK_SPLIT")) : 1;
        const double out_mb = (double)a.M * a.N * 4.0 / 1048576.0;
        sk = long_k ? ceil_div(1024, tiles) : 1;
        while (sk > 1 && (nk / sk < 16 || sk * out_mb > 24.0)) --sk;
      }
      // a fresh (non-accumulated) output has to be cleared by a launch of its own before the slices can add into it: only worth
      // it for a long K loop in a launch that would otherwise leave the chip underfilled
      if (sk > 1 && !a.accumulate && (crowded || nk < 16)) sk = 1;
      if (sk > 1 && !a.accumulate && a.ldc != a.N) sk = 1;
      sks[i] = sk;
    }
    // pass 2: the launch as a whole.  The chip holds SLOTS workgroups of this kernel at once; a launch of 1.x times that runs a second,
    // mostly empty round.  While the launch sits between one and two rounds, the most finely split problems give slices back.
    {
      static const int slots = getenv("MMDA_GEMM_SLOTS") ? atoi(getenv("MMDA_GEMM_SLOTS")) : (T == 64 ? 1024 : 512);
      auto total = [&]() { int64_t t = 0; for (int i : order) t += (int64_t)tiles_of(args[i]) * sks[i]; return t; };
      int64_t tot = total();
      while (slots > 0 && tot > slots && tot < 2 * (int64_t)slots) {
        int best = -1;
        for (int i : order) if (sks[i] > 1 && (best < 0 || sks[i] > sks[best] || (sks[i] == sks[best] && tiles_of(args[i]) > tiles_of(args[best])))) best = i;
        if (best < 0) break;
        --sks[best];
        tot = total();
      }
    }
    for (int i : order) {
      const mmda_gemm_bf16_args& a = args[i];
      if (G.n == GROUP_MAX) { int rc = flush(); if (rc) return rc; }
      const int k = G.n++;
      G.p[k] = a;
      const int Ne = a.N + (a.bias_grad ? 1 : 0);
      G.tile[k] = T;
      G.tx[k] = ceil_div(Ne, T); G.ty[k] = ceil_div(a.M, T);
      const int tiles = G.tx[k] * G.ty[k];
      const int sk = sks[i];
      if (sk > 1 && !a.accumulate && hipMemsetAsync(a.C, 0, sizeof(float) * (size_t)a.M * a.N, s) != hipSuccess) return MMDA_ELAUNCH;
      G.splitk[k] = sk;
      G.start[k] = blocks;
      blocks += tiles * sk;
    }
    int rc = flush();
    if (rc) return rc;
  }
  return MMDA_OK;
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
