// Bidirectional LSTM recurrence for gfx950 (reference: nn.LSTM(bidirectional=True) on packed sequences,
// models.py:48-55 driven by extract_features models.py:163-180; PyTorch gate order i,f,g,o).
//
// Decomposition ("batch-parallel, time-persistent"): one workgroup owns 16 samples of one direction of one modality
// for ALL time steps, so the serial chain never crosses a workgroup: no grid barrier, no inter-CU hand-off.  Per
// step the workgroup computes gates(16 x 4H) = h_{t-1}(16 x H) * W_hh^T on the matrix cores:
//   - h_{t-1} lives in LDS (bf16 or fp32), double-buffered; c_t and an fp32 copy of h_t live in registers
//   - W_hh is streamed from L2 in pre-packed MFMA-fragment order (one contiguous 1 KiB wave-load per fragment)
//   - a wave owns hidden tiles (16 units) and computes all four gates of its units, so the cell update is lane-local
// The input-to-hidden products for all T are done beforehand as one time-batched GEMM (gemm.hip) into `gates`;
// this kernel adds them, applies the nonlinearities and overwrites `gates` with the activated values (the stash
// the backward kernel needs).  Variable lengths: sample b is updated only while t < len_b; the reverse direction
// walks t = T-1..0 from a zero state, so it effectively starts at len_b-1 (packed-sequence semantics).
#include "common.h"
#include "convert_tile.h"
#include "transpose_tile.h"

namespace {

constexpr int NW = 4;          // waves per workgroup
constexpr int MAXDESC = 4;

struct LstmLaunch {
  mmda_lstm_desc d[MAXDESC];
  int n, B, T, nbt;            // nbt = ceil(B/16) batch tiles
  const int32_t* lengths;
};

__device__ __forceinline__ int pad16(int h) { return (h + 15) & ~15; }
__device__ __forceinline__ int pad32(int h) { return (h + 31) & ~31; }

// ------------------------------------------------------------------------------------------------ packing
// bf16 forward : [(ht*4+g)*KS + ks][lane][8]   k = ks*32 + 8*(lane>>4) + j, n = ht*16 + (lane&15), row = g*H+n
// bf16 backward: [ht*KSB + ks][lane][8]        kk = ks*32 + 8*(lane>>4) + j over padded gate rows g*Hp+jj
// f32  forward : [(ht*4+g)*KG + kg][lane][4]   k = kg*16 + 4*i + (lane>>4)
// f32  backward: [ht*KGB + kg][lane][4]        kk = kg*16 + 4*i + (lane>>4)
template <int MODE>
__global__ void pack_kernel(int H, const float* __restrict__ W, void* outF, void* outB) {
  const int Hp = pad16(H), nHT = Hp / 16;
  const int per = (MODE == MMDA_BF16) ? 8 : 4;
  const int kspan = (MODE == MMDA_BF16) ? 32 : 16;
  const int KS = (MODE == MMDA_BF16) ? pad32(H) / 32 : Hp / 16;
  const int KSB = 4 * Hp / kspan;
  int64_t nF = (int64_t)nHT * 4 * KS * 64 * per;
  int64_t nB = (int64_t)nHT * KSB * 64 * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nF + nB; i += (int64_t)gridDim.x * blockDim.x) {
    bool bwd = i >= nF;
    int64_t e = bwd ? i - nF : i;
    int j = e % per;
    int lane = (e / per) % 64;
    int64_t frag = e / (per * 64);
    float v = 0.f;
    int koff = (MODE == MMDA_BF16) ? 8 * (lane >> 4) + j : 4 * j + (lane >> 4);
    if (!bwd) {
      int ks = frag % KS;
      int tg = frag / KS;
      int g = tg & 3, ht = tg >> 2;
      int n = ht * 16 + (lane & 15), k = ks * kspan + koff;
      if (n < H && k < H) v = W[(int64_t)(g * H + n) * H + k];
    } else {
      int ks = frag % KSB;
      int ht = frag / KSB;
      int kk = ks * kspan + koff;
      int g = kk / Hp, jj = kk % Hp;
      int n = ht * 16 + (lane & 15);
      if (jj < H && n < H) v = W[(int64_t)(g * H + jj) * H + n];
    }
    if (MODE == MMDA_BF16) {
      unsigned short* o = reinterpret_cast<unsigned short*>(bwd ? outB : outF);
      if (o) o[e] = f2bf(v);
    } else {
      float* o = reinterpret_cast<float*>(bwd ? outB : outF);
      if (o) o[e] = v;
    }
  }
}

struct PackMulti { int H[16]; const float* W[16]; void* F[16]; void* Bk[16]; void* Ck[16]; int start[17]; int n; };
template <int MODE>
__device__ __forceinline__ void pack_block(const PackMulti& P, int block) {
  // block -> matrix through the prefix table.  One thread per (fragment, lane): it gathers the lane's 8 bf16 (4 fp32) values
  // and writes them with one 16-byte store; loads are unconditional from clamped addresses (values outside H are zeroed after).
  int i = 0;
#pragma unroll
  for (int k = 1; k < 16; ++k)
    if (k < P.n && block >= P.start[k]) i = k;
  const int H = P.H[i];
  const float* __restrict__ W = P.W[i];
  const int Hp = pad16(H), nHT = Hp / 16;
  constexpr int per = (MODE == MMDA_BF16) ? 8 : 4;
  const int kspan = (MODE == MMDA_BF16) ? 32 : 16;
  const int KS = (MODE == MMDA_BF16) ? pad32(H) / 32 : Hp / 16;
  const int KSB = 4 * Hp / kspan;
  const int64_t gF = (int64_t)nHT * 4 * KS * 64, gB = (int64_t)nHT * KSB * 64;        // (fragment, lane) pairs per packing
  const int64_t gC = (MODE == MMDA_BF16 && P.Ck[i]) ? (int64_t)nHT * nHT * 2 * 64 : 0;
  const int64_t q0 = (int64_t)(block - P.start[i]) * blockDim.x + threadIdx.x;
  if (q0 >= gF + gB + gC) return;
  float v[per];
  void* dst;
  int64_t q;
  if (q0 >= gF + gB) {
    // cluster-backward packing [(ht*nHT + nt)*2 + ks2][lane][8]: k = gate rows of hidden tile ht, n = hidden tile nt
    q = q0 - gF - gB;
    dst = P.Ck[i];
    const int lane = (int)(q % 64);
    const int64_t frag = q / 64;
    const int ks2 = (int)(frag & 1), nt = (int)((frag >> 1) % nHT), ht = (int)((frag >> 1) / nHT);
    const int n = nt * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < per; ++j) {
      const int kk = ks2 * 32 + 8 * (lane >> 4) + j;
      const int g = kk >> 4, unit = ht * 16 + (kk & 15);
      const float w = W[(int64_t)(g * H + min(unit, H - 1)) * H + min(n, H - 1)];
      v[j] = (unit < H && n < H) ? w : 0.f;
    }
  } else {
    const bool bwd = q0 >= gF;
    dst = bwd ? P.Bk[i] : P.F[i];
    if (!dst) return;                                  // this packing is not wanted
    q = bwd ? q0 - gF : q0;
    const int lane = (int)(q % 64);
    const int64_t frag = q / 64;
    const int n16 = lane & 15;
#pragma unroll
    for (int j = 0; j < per; ++j) {
      const int koff = (MODE == MMDA_BF16) ? 8 * (lane >> 4) + j : 4 * j + (lane >> 4);
      if (!bwd) {
        const int ks = (int)(frag % KS), tg = (int)(frag / KS), g = tg & 3, ht = tg >> 2;
        const int n = ht * 16 + n16, k = ks * kspan + koff;
        const float w = W[(int64_t)(g * H + min(n, H - 1)) * H + min(k, H - 1)];
        v[j] = (n < H && k < H) ? w : 0.f;
      } else {
        const int ks = (int)(frag % KSB), ht = (int)(frag / KSB), kk = ks * kspan + koff;
        const int g = kk / Hp, jj = kk % Hp, n = ht * 16 + n16;
        const float w = W[(int64_t)(g * H + min(jj, H - 1)) * H + min(n, H - 1)];
        v[j] = (jj < H && n < H) ? w : 0.f;
      }
    }
  }
  if (MODE == MMDA_BF16) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (unsigned)f2bf(v[2 * j]) | ((unsigned)f2bf(v[2 * j + 1]) << 16);
    reinterpret_cast<u32x4_t*>(dst)[q] = o;
  } else {
    reinterpret_cast<float4*>(dst)[q] = float4{v[0], v[1], v[2], v[3]};
  }
}

template <int MODE>
__global__ void pack_multi_kernel(PackMulti P) { pack_block<MODE>(P, (int)blockIdx.x); }

// The twelve W_hh packings of a step AND its first bf16 conversions (W_ih of both layers, the layer-1 inputs incl. the embedding
// lookup) in ONE launch: blocks [0, pack_blocks) pack, the rest convert.  Both only depend on the step's inputs and weights, and as
// two launches on two streams they cost the main stream a fork and a cross-stream wait in front of the first recurrent kernel.
// Blocks past the conversions transpose fp32 matrices (the fusion block's K-major weight copies for the backward pass).
__global__ __launch_bounds__(256) void pack_convert_kernel(PackMulti P, ConvLaunch L, TrLaunch R, int pack_blocks, int conv_end) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][66];
  static_assert(sizeof(tile) >= sizeof(float) * 32 * 33, "the transpose tile is laid over the conversion tile");
  if ((int)blockIdx.x < pack_blocks) pack_block<MMDA_BF16>(P, (int)blockIdx.x);
  else if ((int)blockIdx.x < conv_end) convert_block(L, (int)blockIdx.x - pack_blocks, tile);
  else transpose_block(R, (int)blockIdx.x - conv_end, reinterpret_cast<float (*)[33]>(&tile[0][0]));
}

template <int MODE> __device__ __forceinline__ float sig_(float x) { return MODE == MMDA_BF16 ? sigmoid_fast(x) : sigmoidf_(x); }
template <int MODE> __device__ __forceinline__ float tanh_(float x) { return MODE == MMDA_BF16 ? tanh_fast(x) : tanhf_(x); }

// ------------------------------------------------------------------------------------------------ forward
// CELL = MMDA_CELL_GRU: nn.GRU (reference models.py:39) on the same machinery.  The caller lays the GRU's three gate blocks out
// in the four LSTM slots: W_ih rows [r; z; n; 0], W_hh rows [r; z; 0; n], so slot 2 of `gates` holds x W_in^T + b_in, slot 3
// of `gates` holds b_hn and the matrix cores add h W_hn^T to it:  r = s(s0), z = s(s1), q = s3, n = tanh(s2 + r q),
// h' = (1 - z) n + z h.  Stash: gates <- [r, z, n, q], cstash <- h'.
template <int MODE, int MAXT, int CELL>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_kernel(LstmLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int per_mod = 2 * L.nbt;
  const int mod = blockIdx.x / per_mod;
  const int rem = blockIdx.x % per_mod;
  const int dir = rem / L.nbt, bt = rem % L.nbt;
  const mmda_lstm_desc& D = L.d[mod];
  const int H = D.H, Hp = pad16(H), nHT = Hp / 16, Kp = pad32(H);
  const int B = L.B, T = L.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  // LDS h buffers
  const int ld = (MODE == MMDA_BF16) ? Kp + 8 : Kp + 2;
  unsigned short* hb16 = reinterpret_cast<unsigned short*>(smem);
  float* hb32 = reinterpret_cast<float*>(smem);
  {
    int total = 2 * 16 * ld;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      if (MODE == MMDA_BF16) hb16[i] = 0; else hb32[i] = 0.f;
    }
  }
  float c_reg[MAXT][4], h_reg[MAXT][4];
  int len_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int b = bt * 16 + fq * 4 + r;
    len_r[r] = b < B ? L.lengths[b] : 0;
  }
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) { c_reg[ti][r] = 0.f; h_reg[ti][r] = 0.f; }
  __syncthreads();

  const int KS = (MODE == MMDA_BF16) ? Kp / 32 : Hp / 16;
  const int G4 = 4 * H;
  int cur = 0;
  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) {
      const int ht = wave + NW * ti;
      if (ht >= nHT) continue;
      const int j = ht * 16 + fr;
      // prefetch this step's input-to-hidden pre-activations (independent of h, so issued before the MFMAs)
      float pre[4][4];
      int64_t gbase[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = bt * 16 + fq * 4 + r;
        bool act = (j < H) && (t < len_r[r]);
        gbase[r] = (((int64_t)t * B + b) * 2 + dir) * G4 + j;
        // unconditional loads from clamped addresses + select (a load under a lane-dependent branch serialises on vmcnt(0))
        int64_t gsafe = (((int64_t)t * B + min(b, B - 1)) * 2 + dir) * G4 + min(j, H - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) { float v = D.gates[gsafe + g * H]; pre[g][r] = act ? v : 0.f; }
      }
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (MODE == MMDA_BF16) {
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(D.wpack[dir]) + (int64_t)(ht * 4) * KS * 64 + lane;
        const unsigned short* hrow = hb16 + (cur * 16 + fr) * ld + fq * 8;
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + ks * 32);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x8 bw = wp[(int64_t)(g * KS + ks) * 64];
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw, acc[g], 0, 0, 0);
          }
        }
      } else {
        const float4* wp = reinterpret_cast<const float4*>(D.wpack[dir]) + (int64_t)(ht * 4) * KS * 64 + lane;
        const float* hrow = hb32 + (cur * 16 + fr) * ld + fq;
#pragma unroll 2
        for (int kg = 0; kg < KS; ++kg) {
          float a0 = hrow[kg * 16 + 0], a1 = hrow[kg * 16 + 4], a2 = hrow[kg * 16 + 8], a3 = hrow[kg * 16 + 12];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 bw = wp[(int64_t)(g * KS + kg) * 64];
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bw.x, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bw.y, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bw.z, acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bw.w, acc[g], 0, 0, 0);
          }
        }
      }
      // lane-local cell update: accumulator element r is sample fq*4+r, hidden unit j
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = bt * 16 + fq * 4 + r;
        bool inb = (b < B) && (j < H);
        bool act = inb && (t < len_r[r]);
        if (act) {
          float gi = sig_<MODE>(acc[0][r] + pre[0][r]);
          float gf = sig_<MODE>(acc[1][r] + pre[1][r]);
          float gg, go, cn, hn;
          if (CELL == MMDA_CELL_GRU) {
            go = acc[3][r] + pre[3][r];
            gg = tanh_<MODE>(acc[2][r] + pre[2][r] + gi * go);
            hn = (1.f - gf) * gg + gf * h_reg[ti][r];
            cn = hn;
          } else {
            gg = tanh_<MODE>(acc[2][r] + pre[2][r]);
            go = sig_<MODE>(acc[3][r] + pre[3][r]);
            cn = gf * c_reg[ti][r] + gi * gg;
            hn = go * tanh_<MODE>(cn);
          }
          c_reg[ti][r] = cn;
          h_reg[ti][r] = hn;
          D.gates[gbase[r]] = gi;
          D.gates[gbase[r] + H] = gf;
          D.gates[gbase[r] + 2 * H] = gg;
          D.gates[gbase[r] + 3 * H] = go;
          D.cstash[(((int64_t)t * B + b) * 2 + dir) * H + j] = cn;
          D.hseq[((int64_t)t * B + b) * 2 * H + dir * H + j] = hn;
        } else if (inb) {
          D.hseq[((int64_t)t * B + b) * 2 * H + dir * H + j] = 0.f;   // pad_packed_sequence zero fill
        }
        int s = fq * 4 + r;
        if (MODE == MMDA_BF16) hb16[((cur ^ 1) * 16 + s) * ld + j] = f2bf(h_reg[ti][r]);
        else hb32[((cur ^ 1) * 16 + s) * ld + j] = h_reg[ti][r];
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  // final hidden state straight into the utterance layout [h1_fwd, h2_fwd, h1_bwd, h2_bwd] (models.py:203)
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti) {
    const int ht = wave + NW * ti;
    if (ht >= nHT) continue;
    const int j = ht * 16 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int b = bt * 16 + fq * 4 + r;
      if (b < B && j < H) D.utt[(int64_t)b * 4 * H + (dir * 2 + D.layer) * H + j] = h_reg[ti][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Walks time in the reverse of the forward order.  Per step: (1) lane-local gate gradients from dh, dc and the
// stash, written in place over `gates` (fp32, for the weight/input gradient GEMMs) and to LDS as the MFMA A operand;
// (2) dh_{t-1}(16 x H) = dG(16 x 4H) * W_hh on the matrix cores with W_hh streamed in the backward packing.
// CELL = MMDA_CELL_GRU (stash [r, z, n, q], cstash = h): with dh the total gradient of h_t,
//   dn = dh (1 - z), dz = dh (h_{t-1} - n), dpre_n = dn (1 - n^2), dq = dpre_n r, dr = dpre_n q;
//   gates <- [dr r (1 - r), dz z (1 - z), dpre_n, dq]; the direct path dh z to h_{t-1} rides in the `dc` carry.
template <int MODE, int MAXT, int CELL>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_kernel(LstmLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int per_mod = 2 * L.nbt;
  const int mod = blockIdx.x / per_mod;
  const int rem = blockIdx.x % per_mod;
  const int dir = rem / L.nbt, bt = rem % L.nbt;
  const mmda_lstm_desc& D = L.d[mod];
  const int H = D.H, Hp = pad16(H), nHT = Hp / 16;
  const int B = L.B, T = L.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int KW = 4 * Hp;                                     // padded contraction length (gate rows)
  const int ld = (MODE == MMDA_BF16) ? KW + 8 : KW + 2;
  unsigned short* g16 = reinterpret_cast<unsigned short*>(smem);
  float* g32 = reinterpret_cast<float*>(smem);
  const int KSB = (MODE == MMDA_BF16) ? KW / 32 : KW / 16;
  const int G4 = 4 * H;

  float dh_rec[MAXT][4], dc[MAXT][4];
  int len_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int b = bt * 16 + fq * 4 + r;
    len_r[r] = b < B ? L.lengths[b] : 0;
  }
#pragma unroll
  for (int ti = 0; ti < MAXT; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) { dh_rec[ti][r] = 0.f; dc[ti][r] = 0.f; }

  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : T - 1 - step;
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) {
      const int ht = wave + NW * ti;
      if (ht >= nHT) continue;
      const int j = ht * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = bt * 16 + fq * 4 + r;
        bool inb = (b < B) && (j < H);
        bool act = inb && (t < len_r[r]);
        float dp[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t gb = (((int64_t)t * B + b) * 2 + dir) * G4 + j;
        // unconditional stash loads from clamped addresses (no lane-dependent branch around a load)
        const int bc = min(b, B - 1), jc = min(j, H - 1);
        const int64_t gs = (((int64_t)t * B + bc) * 2 + dir) * G4 + jc;
        const float l_gi = D.gates[gs], l_gf = D.gates[gs + H], l_gg = D.gates[gs + 2 * H], l_go = D.gates[gs + 3 * H];
        const float l_ct = D.cstash[(((int64_t)t * B + bc) * 2 + dir) * H + jc];
        const int tp = dir ? t + 1 : t - 1;
        const float l_cp = D.cstash[(((int64_t)min(max(tp, 0), T - 1) * B + bc) * 2 + dir) * H + jc];
        float l_dh = 0.f;
        if (D.d_hseq) l_dh = D.d_hseq[((int64_t)t * B + bc) * 2 * H + dir * H + jc];
        const float l_ut = D.utt[(int64_t)bc * 4 * H + (dir * 2 + D.layer) * H + jc];
        if (act) {
          float gi = l_gi, gf = l_gf, gg = l_gg, go = l_go;
          float ct = l_ct;
          float cp = (tp >= 0 && tp < len_r[r]) ? l_cp : 0.f;
          float dh = dh_rec[ti][r] + l_dh;
          bool fin = dir ? (t == 0) : (t == len_r[r] - 1);
          dh += fin ? l_ut : 0.f;
          if (CELL == MMDA_CELL_GRU) {
            dh += dc[ti][r];
            float dpn = dh * (1.f - gf) * (1.f - gg * gg);
            dp[0] = dpn * go * gi * (1.f - gi);
            dp[1] = dh * (cp - gg) * gf * (1.f - gf);
            dp[2] = dpn;
            dp[3] = dpn * gi;
            dc[ti][r] = dh * gf;
          } else {
            float tc = tanh_<MODE>(ct);
            float dct = dc[ti][r] + dh * go * (1.f - tc * tc);
            dp[0] = dct * gg * gi * (1.f - gi);
            dp[1] = dct * cp * gf * (1.f - gf);
            dp[2] = dct * gi * (1.f - gg * gg);
            dp[3] = dh * tc * go * (1.f - go);
            dc[ti][r] = dct * gf;
          }
        }
        if (inb) {
#pragma unroll
          for (int g = 0; g < 4; ++g) D.gates[gb + g * H] = dp[g];
        }
        int s = fq * 4 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (MODE == MMDA_BF16) g16[s * ld + g * Hp + j] = f2bf(dp[g]);
          else g32[s * ld + g * Hp + j] = dp[g];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) {
      const int ht = wave + NW * ti;
      if (ht >= nHT) continue;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (MODE == MMDA_BF16) {
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(D.wpack[dir]) + (int64_t)ht * KSB * 64 + lane;
        const unsigned short* arow = g16 + fr * ld + fq * 8;
#pragma unroll 4
        for (int ks = 0; ks < KSB; ++ks) {
          bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
          bf16x8 bw = wp[(int64_t)ks * 64];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw, acc, 0, 0, 0);
        }
      } else {
        const float4* wp = reinterpret_cast<const float4*>(D.wpack[dir]) + (int64_t)ht * KSB * 64 + lane;
        const float* arow = g32 + fr * ld + fq;
#pragma unroll 2
        for (int kg = 0; kg < KSB; ++kg) {
          float4 bw = wp[(int64_t)kg * 64];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[kg * 16 + 0], bw.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[kg * 16 + 4], bw.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[kg * 16 + 8], bw.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[kg * 16 + 12], bw.w, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh_rec[ti][r] = acc[r];
    }
    __syncthreads();
  }
}

}  // namespace
int mmda_lstm_cluster_launch(int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream, bool bwd,
                             int* used);   // lstm_cluster.hip
namespace {

int pick_maxt(int n, const mmda_lstm_desc* d) {
  int mx = 0;
  for (int i = 0; i < n; ++i) mx = d[i].H > mx ? d[i].H : mx;
  int nht = round_up(mx, 16) / 16;
  return ceil_div(nht, NW);
}

template <int MODE, bool BWD, int CELL>
int launch(int maxt, const LstmLaunch& L, size_t lds, hipStream_t s) {
  dim3 grid(L.n * 2 * L.nbt), block(NW * 64);
#define LAUNCH_T(MT)                                                                                      \
  do {                                                                                                    \
    auto kfn = BWD ? lstm_bwd_kernel<MODE, MT, CELL> : lstm_fwd_kernel<MODE, MT, CELL>;                   \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds) != hipSuccess) { (void)hipGetLastError(); }                         \
    hipLaunchKernelGGL(kfn, grid, block, lds, s, L);                                                      \
  } while (0)
  if (maxt <= 1) LAUNCH_T(1);
  else if (maxt <= 2) LAUNCH_T(2);
  else if (maxt <= 3) LAUNCH_T(3);
  else if (maxt <= 5) LAUNCH_T(5);
  else if (maxt <= 8) LAUNCH_T(8);
  else return MMDA_EINVAL;
#undef LAUNCH_T
  return MMDA_OK;
}

int lstm_common(int mode, int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream, bool bwd) {
  if (n <= 0 || n > MAXDESC || !descs || B <= 0 || T < 0 || !lengths) return MMDA_EINVAL;
  if (mode != MMDA_F32 && mode != MMDA_BF16) return MMDA_EINVAL;
  if (T == 0) return MMDA_OK;
  LstmLaunch L;
  L.n = n; L.B = B; L.T = T; L.nbt = ceil_div(B, 16); L.lengths = lengths;
  int maxH = 0;
  for (int i = 0; i < n; ++i) {
    const mmda_lstm_desc& d = descs[i];
    if (d.H <= 0 || d.H > 512 || !d.gates || !d.cstash || !d.wpack[0] || !d.wpack[1] || !d.utt) return MMDA_EINVAL;
    if (!bwd && !d.hseq) return MMDA_EINVAL;
    if (d.layer != 0 && d.layer != 1) return MMDA_EINVAL;
    if ((d.cell != MMDA_CELL_LSTM && d.cell != MMDA_CELL_GRU) || d.cell != descs[0].cell) return MMDA_EINVAL;
    L.d[i] = d;
    maxH = d.H > maxH ? d.H : maxH;
  }
  for (int i = n; i < MAXDESC; ++i) L.d[i] = descs[0];
  if (mode == MMDA_BF16) {      // weights resident on chip when every descriptor brought an exchange buffer
    int used = 0;
    int rc = mmda_lstm_cluster_launch(n, descs, B, T, lengths, stream, bwd, &used);
    if (rc != MMDA_OK) return rc;
    if (used) return MMDA_OK;
  }
  for (int i = 0; i < n; ++i)
    if (descs[i].gate_minor) return MMDA_EINVAL;       // the streaming kernels only know torch's [dir][gate][unit] column order
  int Hp = round_up(maxH, 16), Kp = round_up(maxH, 32);
  size_t lds;
  if (!bwd) lds = (mode == MMDA_BF16) ? (size_t)2 * 16 * (Kp + 8) * 2 : (size_t)2 * 16 * (Kp + 2) * 4;
  else lds = (mode == MMDA_BF16) ? (size_t)16 * (4 * Hp + 8) * 2 : (size_t)16 * (4 * Hp + 2) * 4;
  if (lds > 160 * 1024) return MMDA_EINVAL;
  int maxt = pick_maxt(n, descs);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const bool gru = descs[0].cell == MMDA_CELL_GRU;
#define PICK(MODE_)                                                                                              \
  (gru ? (bwd ? launch<MODE_, true, MMDA_CELL_GRU>(maxt, L, lds, s) : launch<MODE_, false, MMDA_CELL_GRU>(maxt, L, lds, s))  \
       : (bwd ? launch<MODE_, true, MMDA_CELL_LSTM>(maxt, L, lds, s) : launch<MODE_, false, MMDA_CELL_LSTM>(maxt, L, lds, s)))
  if (mode == MMDA_BF16) rc = PICK(MMDA_BF16);
  else rc = PICK(MMDA_F32);
#undef PICK
  if (rc != MMDA_OK) return rc;
  MMDA_CHECK_LAUNCH(bwd ? "mmda_lstm_bwd" : "mmda_lstm_fwd");
  return MMDA_OK;
}

}  // namespace

extern "C" int64_t mmda_lstm_packed_bytes(int mode, int H, int backward) {
  if (H <= 0 || H > 512) return MMDA_EINVAL;
  int Hp = round_up(H, 16), nHT = Hp / 16;
  if (backward == 2) return mode == MMDA_BF16 ? (int64_t)nHT * nHT * 2 * 64 * 8 * 2 : MMDA_EINVAL;
  if (mode == MMDA_BF16) {
    int KS = round_up(H, 32) / 32, KSB = 4 * Hp / 32;
    return backward ? (int64_t)nHT * KSB * 64 * 8 * 2 : (int64_t)nHT * 4 * KS * 64 * 8 * 2;
  } else if (mode == MMDA_F32) {
    int KG = Hp / 16, KGB = 4 * Hp / 16;
    return backward ? (int64_t)nHT * KGB * 64 * 4 * 4 : (int64_t)nHT * 4 * KG * 64 * 4 * 4;
  }
  return MMDA_EINVAL;
}

extern "C" int mmda_lstm_pack_whh(int mode, int H, const float* whh, void* packed_fwd, void* packed_bwd, void* stream) {
  if (!whh || H <= 0 || H > 512 || (!packed_fwd && !packed_bwd)) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int64_t total = (mmda_lstm_packed_bytes(mode, H, 0) + mmda_lstm_packed_bytes(mode, H, 1)) / (mode == MMDA_BF16 ? 2 : 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (mode == MMDA_BF16) hipLaunchKernelGGL(pack_kernel<MMDA_BF16>, dim3(blocks), dim3(256), 0, s, H, whh, packed_fwd, packed_bwd);
  else if (mode == MMDA_F32) hipLaunchKernelGGL(pack_kernel<MMDA_F32>, dim3(blocks), dim3(256), 0, s, H, whh, packed_fwd, packed_bwd);
  else return MMDA_EINVAL;
  MMDA_CHECK_LAUNCH("mmda_lstm_pack_whh");
  return MMDA_OK;
}

namespace {
int pack_build(int mode, int n, const int* H, const float* const* whh, void* const* packed_fwd, void* const* packed_bwd,
               void* const* packed_c, PackMulti& P, int& blocks) {
  if (n <= 0 || n > 16 || !H || !whh || !packed_fwd || !packed_bwd || (mode != MMDA_BF16 && mode != MMDA_F32)) return MMDA_EINVAL;
  P.n = n;
  blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (H[i] <= 0 || H[i] > 512 || !whh[i] || (!packed_fwd[i] && !packed_bwd[i])) return MMDA_EINVAL;
    P.H[i] = H[i]; P.W[i] = whh[i]; P.F[i] = packed_fwd[i]; P.Bk[i] = packed_bwd[i];
    P.Ck[i] = (packed_c && mode == MMDA_BF16) ? packed_c[i] : nullptr;
    P.start[i] = blocks;
    // one thread per 16 bytes of packed output
    int64_t total = (mmda_lstm_packed_bytes(mode, H[i], 0) + mmda_lstm_packed_bytes(mode, H[i], 1)) / 16;
    if (P.Ck[i]) total += mmda_lstm_packed_bytes(mode, H[i], 2) / 16;
    blocks += (int)((total + 255) / 256);
  }
  for (int i = n; i < 16; ++i) { P.H[i] = P.H[0]; P.W[i] = P.W[0]; P.F[i] = P.F[0]; P.Bk[i] = P.Bk[0]; P.Ck[i] = P.Ck[0]; }
  for (int i = n; i <= 16; ++i) P.start[i] = blocks;
  return MMDA_OK;
}
}  // namespace

extern "C" int mmda_lstm_pack_convert_transpose(int n, const int* H, const float* const* whh, void* const* packed_fwd,
                                                void* const* packed_bwd, void* const* packed_c, const mmda_convert_job* jobs, int njobs,
                                                const mmda_transpose_job* tjobs, int ntjobs, void* stream) {
  if (!jobs || njobs < 0 || njobs > CONV_MAX) return MMDA_EINVAL;
  PackMulti P;
  int pblocks = 0, cblocks = 0, tblocks = 0;
  int rc = pack_build(MMDA_BF16, n, H, whh, packed_fwd, packed_bwd, packed_c, P, pblocks);
  if (rc) return rc;
  ConvLaunch L;
  rc = conv_build(jobs, njobs, L, cblocks);
  if (rc) return rc;
  TrLaunch R;
  rc = tr_build(tjobs, ntjobs, R, tblocks);
  if (rc) return rc;
  if (pblocks + cblocks + tblocks == 0) return MMDA_OK;
  hipLaunchKernelGGL(pack_convert_kernel, dim3(pblocks + cblocks + tblocks), dim3(256), 0, (hipStream_t)stream, P, L, R, pblocks,
                     pblocks + cblocks);
  MMDA_CHECK_LAUNCH("mmda_lstm_pack_convert_transpose");
  return MMDA_OK;
}

extern "C" int mmda_lstm_pack_whh_and_convert(int n, const int* H, const float* const* whh, void* const* packed_fwd, void* const* packed_bwd,
                                              void* const* packed_c, const mmda_convert_job* jobs, int njobs, void* stream) {
  return mmda_lstm_pack_convert_transpose(n, H, whh, packed_fwd, packed_bwd, packed_c, jobs, njobs, nullptr, 0, stream);
}

extern "C" int mmda_lstm_pack_whh_multi(int mode, int n, const int* H, const float* const* whh, void* const* packed_fwd,
                                        void* const* packed_bwd, void* const* packed_c, void* stream) {
  PackMulti P;
  int blocks = 0;
  const int rc0 = pack_build(mode, n, H, whh, packed_fwd, packed_bwd, packed_c, P, blocks);
  if (rc0) return rc0;
  hipStream_t s = (hipStream_t)stream;
  if (mode == MMDA_BF16) hipLaunchKernelGGL(pack_multi_kernel<MMDA_BF16>, dim3(blocks), dim3(256), 0, s, P);
  else hipLaunchKernelGGL(pack_multi_kernel<MMDA_F32>, dim3(blocks), dim3(256), 0, s, P);
  MMDA_CHECK_LAUNCH("mmda_lstm_pack_whh_multi");
  return MMDA_OK;
}

extern "C" int mmda_lstm_pack_whh_cluster(int H, const float* whh, void* packed_c, void* stream) {
  // single-matrix form on top of the multi kernel: the forward / streaming-backward packings are not wanted (NULL: their threads
  // exit at once), only the cluster segment is written
  if (!whh || !packed_c || H <= 0 || H > 512) return MMDA_EINVAL;
  PackMulti P;
  P.n = 1;
  for (int i = 0; i < 16; ++i) { P.H[i] = H; P.W[i] = whh; P.F[i] = nullptr; P.Bk[i] = nullptr; P.Ck[i] = packed_c; }
  const int64_t total = (mmda_lstm_packed_bytes(MMDA_BF16, H, 0) + mmda_lstm_packed_bytes(MMDA_BF16, H, 1) +
                         mmda_lstm_packed_bytes(MMDA_BF16, H, 2)) / 16;
  const int blocks = (int)((total + 255) / 256);
  P.start[0] = 0;
  for (int i = 1; i <= 16; ++i) P.start[i] = blocks;
  hipLaunchKernelGGL(pack_multi_kernel<MMDA_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
  MMDA_CHECK_LAUNCH("mmda_lstm_pack_whh_cluster");
  return MMDA_OK;
}

// ------------------------------------------------------------------------------------------------ GRU <-> four-slot layout
namespace {
struct GruPad { mmda_gru_pad_job j[MMDA_GRU_PAD_MAX]; int start[MMDA_GRU_PAD_MAX + 1]; int n; };

// One thread per element of the padded tensors of one layer: [W_ih (8H,D) | W_hh fwd (4H,H) | W_hh rev (4H,H) | bias (8H)].
// UNPAD = false: padded <- torch layout (zero rows where a slot has no torch gate).
// UNPAD = true : torch-layout gradient += padded gradient, padded gradient <- 0 (so the next step accumulates from zero).
template <bool UNPAD>
__global__ void gru_pad_kernel(GruPad P) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < MMDA_GRU_PAD_MAX; ++k)
    if (k < P.n && (int)blockIdx.x >= P.start[k]) i = k;
  const mmda_gru_pad_job& J = P.j[i];
  const int H = J.H, D = J.D;
  const int64_t nih = (int64_t)8 * H * D, nhh = (int64_t)4 * H * H, nb = 8 * H;
  int64_t e = (int64_t)(blockIdx.x - P.start[i]) * blockDim.x + threadIdx.x;
  if (e < nih) {
    const int row = (int)(e / D), col = (int)(e % D);
    const int dir = row / (4 * H), slot = (row % (4 * H)) / H, u = row % H;
    float* t = J.w_ih[dir] + ((int64_t)slot * H + u) * D + col;            // slots 0..2 = torch's r, z, n
    if (!UNPAD) J.pw_ih[e] = slot < 3 ? *t : 0.f;
    else { float v = J.pw_ih[e]; J.pw_ih[e] = 0.f; if (slot < 3) *t += v; }
    return;
  }
  e -= nih;
  if (e < 2 * nhh) {
    const int dir = (int)(e / nhh);
    const int64_t e3 = e % nhh;
    const int row = (int)(e3 / H), col = (int)(e3 % H), slot = row / H, u = row % H;
    const int tg = slot == 3 ? 2 : slot;                                    // slot 3 = torch's n; slot 2 has no W_hh rows
    float* t = J.w_hh[dir] + ((int64_t)tg * H + u) * H + col;
    if (!UNPAD) J.pw_hh[dir][e3] = slot != 2 ? *t : 0.f;
    else { float v = J.pw_hh[dir][e3]; J.pw_hh[dir][e3] = 0.f; if (slot != 2) *t += v; }
    return;
  }
  e -= 2 * nhh;
  if (e < nb) {
    const int dir = (int)e / (4 * H), slot = ((int)e % (4 * H)) / H, u = (int)e % H;
    const int tg = slot == 3 ? 2 : slot;
    float* ti = J.b_ih[dir] + slot * H + u;
    float* th = J.b_hh[dir] + tg * H + u;
    if (!UNPAD) {
      J.pb_ih[e] = slot < 3 ? *ti : 0.f;
      J.pb_hh[e] = slot != 2 ? *th : 0.f;
    } else {
      float v = J.pb_ih[e]; J.pb_ih[e] = 0.f;                               // column sums of the four-slot gate gradients
      if (slot < 3) *ti += v;
      if (slot != 2) *th += v;
    }
  }
}

int gru_pad_common(const mmda_gru_pad_job* jobs, int n, void* stream, bool unpad) {
  if (!jobs || n <= 0 || n > MMDA_GRU_PAD_MAX) return MMDA_EINVAL;
  GruPad P;
  P.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const mmda_gru_pad_job& j = jobs[i];
    if (j.H <= 0 || j.H > 512 || j.D <= 0 || !j.pw_ih || !j.pb_ih || (!unpad && !j.pb_hh)) return MMDA_EINVAL;
    for (int d = 0; d < 2; ++d)
      if (!j.w_ih[d] || !j.w_hh[d] || !j.b_ih[d] || !j.b_hh[d] || !j.pw_hh[d]) return MMDA_EINVAL;
    P.j[i] = j;
    P.start[i] = blocks;
    int64_t total = (int64_t)8 * j.H * j.D + (int64_t)8 * j.H * j.H + 8 * j.H;
    blocks += (int)((total + 255) / 256);
  }
  for (int i = n; i < MMDA_GRU_PAD_MAX; ++i) P.j[i] = P.j[0];
  for (int i = n; i <= MMDA_GRU_PAD_MAX; ++i) P.start[i] = blocks;
  if (unpad) hipLaunchKernelGGL(gru_pad_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
  else hipLaunchKernelGGL(gru_pad_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P);
  MMDA_CHECK_LAUNCH(unpad ? "mmda_gru_unpad_grads" : "mmda_gru_pad_params");
  return MMDA_OK;
}
}  // namespace

extern "C" int mmda_gru_pad_params(const mmda_gru_pad_job* jobs, int n, void* stream) { return gru_pad_common(jobs, n, stream, false); }
extern "C" int mmda_gru_unpad_grads(const mmda_gru_pad_job* jobs, int n, void* stream) { return gru_pad_common(jobs, n, stream, true); }

extern "C" int mmda_lstm_fwd(int mode, int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream) {
  return lstm_common(mode, n, descs, B, T, lengths, stream, false);
}
extern "C" int mmda_lstm_bwd(int mode, int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream) {
  return lstm_common(mode, n, descs, B, T, lengths, stream, true);
}
