// biLSTM recurrence with W_hh RESIDENT ON CHIP (bf16 mode) - the fast path behind mmda_lstm_fwd / mmda_lstm_bwd.
//
// lstm.hip re-streams W_hh (720 KB bf16 for the 300-wide text LSTM) from L2 through one CU on every one of the 2*T
// dependent steps; that stream, not the matrix cores, sets its step time.  Here the hidden units of one
// (modality, direction, 32-sample batch group) are split over a small CLUSTER of NC workgroups (one per CU); each
// workgroup keeps the W_hh rows (forward) / columns (backward) of its own hidden tiles in LDS for the whole sequence, so a
// step is: MFMA out of LDS (~0.3 us) + lane-local cell update + one all-gather of the new h (forward, 20 KB) or of the
// gate gradients (backward, 78 KB) among the NC workgroups of the cluster.  Text (H=300): NC=10, visual/acoustic: NC=1
// (no exchange at all).
//
// Exchange protocol (MI355X guide, Guideline 16, form R1 with 8-byte write-through granules):
//   producer: own slice -> 8-byte agent-scope relaxed atomic stores (sc1, write-through) into X[epoch&1]; EVERY storing
//             wave drains (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane stores flag[me] = epoch (agent-scope atomic)
//   consumer: ONE wave polls the other NC-1 flags (relaxed agent-scope loads, lane per producer, s_sleep, BOUNDED spin);
//             workgroup barrier; every load of the handed-off bytes is an 8-byte agent-scope relaxed atomic load (sc1:
//             bypasses this CU's L1, so no acquire fence is needed); results go to LDS
//   epochs are monotonic across steps AND launches (base passed by the host), so flags are never reset; X is double
//   buffered by epoch parity (a producer can only overwrite a buffer two epochs later, after every consumer has signalled
//   the epoch in between).  A timed-out poll sets a sticky abort word, every workgroup leaves its time loop at the next
//   barrier, the grid always drains.  All workgroups of a launch are co-resident: the host caps a launch at 240
//   single-workgroup-per-CU blocks (LDS > 80 KB each) and chunks larger batches over several launches.
#include "common.h"

namespace {

constexpr int GROUP = 32;            // samples per cluster (two 16-row MFMA tiles)
constexpr int MAXD = 4;
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr int MAX_WG_PER_LAUNCH = 240;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

struct CDesc {
  int H, Hp, Kp, KS, KSB, nHT, TPW, NC;
  float* gates; float* cstash; float* hseq; const void* wpack[2]; float* utt; int layer; const float* d_hseq;
  unsigned char* xchg;
  int wg_begin;
};
struct CLaunch {
  CDesc d[MAXD];
  int n, B, T, g0, ng;               // batch groups [g0, g0+ng) of this launch
  const int32_t* lengths;
  unsigned epoch_base;
};

// xchg layout per descriptor: [0,64) abort word | flags: (dir, group, wg) x 64 B | X: (dir, group, parity) x GROUP x XW bf16
__host__ __device__ inline size_t xchg_flags_off() { return 64; }
__host__ __device__ inline size_t xchg_x_off(int ngroups_total, int NC) { return 64 + (size_t)2 * ngroups_total * NC * 64; }
__host__ __device__ inline size_t xchg_bytes(int ngroups_total, int NC, int Hp) {
  return xchg_x_off(ngroups_total, NC) + (size_t)2 * ngroups_total * 2 * GROUP * (4 * Hp) * 2;
}

__device__ __forceinline__ void st_rlx(void* p, u64 v) { __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_rlx(const void* p) { return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(void* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_flag(const void* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// 16-byte write-through (sc1) store / L1-bypassing (sc1) load through a buffer descriptor built from wave-uniform values
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
__device__ __forceinline__ u32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t X_of(unsigned epoch, const __amdgpu_buffer_rsrc_t (&xr)[2]) {
  return (epoch & 1u) ? xr[1] : xr[0];
}

struct Where { int di, dir, grp, me; };
__device__ __forceinline__ Where locate(const CLaunch& L) {
  Where w;
  w.di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < L.n && (int)blockIdx.x >= L.d[i].wg_begin) w.di = i;
  const CDesc& D = L.d[w.di];
  int local = blockIdx.x - D.wg_begin;
  w.dir = local / (L.ng * D.NC);
  int rem = local % (L.ng * D.NC);
  w.grp = L.g0 + rem / D.NC;
  w.me = rem % D.NC;
  return w;
}

// Waits until every other workgroup of the cluster has published `epoch`.  Returns false (uniformly for the workgroup,
// through `lds_ok`) on timeout / abort.  Called by all threads.
__device__ __forceinline__ bool wait_cluster(unsigned char* flags, unsigned char* abort_w, int NC, int me, unsigned epoch,
                                             volatile int* lds_ok) {
  if ((threadIdx.x >> 6) == 0) {
    const int lane = threadIdx.x & 63;
    bool ok = true;
    if (lane < NC && lane != me) {
      unsigned spins = 0;
      while ((int)(ld_flag(flags + (size_t)lane * 64) - epoch) < 0) {
        ++spins;
        if (spins > SPIN_LIMIT || ((spins & 255u) == 0 && ld_flag(abort_w) != 0)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    ok = __all(ok);
    if (lane == 0) {
      *lds_ok = ok ? 1 : 0;
      if (!ok) st_flag(abort_w, 1u);
    }
  }
  __syncthreads();
  return *lds_ok != 0;
}

template <bool FAST> __device__ __forceinline__ float sg(float x) { return sigmoid_fast(x); }
__device__ __forceinline__ float th(float x) { return tanh_fast(x); }

// ------------------------------------------------------------------------------------------------ forward
template <int MAXTW>
__global__ __launch_bounds__(256, 1) void lstm_fwd_cluster_kernel(CLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Where wh = locate(L);
  const CDesc& D = L.d[wh.di];
  const int H = D.H, Hp = D.Hp, Kp = D.Kp, KS = D.KS, nHT = D.nHT, TPW = D.TPW, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir, me = wh.me;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int mt = wave & 1, htl = wave >> 1;           // wave -> (16-sample m-tile, local hidden tile parity)
  const int ld = Kp + 8;
  const int XW = 4 * Hp;
  const int ngt = (B + GROUP - 1) / GROUP;

  uint4* Wl = reinterpret_cast<uint4*>(smem);                                            // TPW*4*KS*64 x 16 B
  unsigned short* hb = reinterpret_cast<unsigned short*>(smem + (size_t)TPW * 4 * KS * 1024);   // 2 x GROUP x ld
  volatile int& lds_ok = *reinterpret_cast<volatile int*>(smem + (size_t)TPW * 4 * KS * 1024 + (size_t)2 * GROUP * ld * 2);

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * 64;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * GROUP * XW * 2;

  // resident weights: the forward packing is [(ht*4+g)*KS + ks][lane] x 16 B; this workgroup owns tiles me*TPW..+TPW
  {
    const uint4* src = reinterpret_cast<const uint4*>(D.wpack[dir]);
    const int per_tile = 4 * KS * 64;
    for (int i = tid; i < TPW * per_tile; i += 256) {
      int lt = i / per_tile, ht = me * TPW + lt;
      Wl[i] = ht < nHT ? src[(size_t)ht * per_tile + (i % per_tile)] : uint4{0, 0, 0, 0};
    }
    for (int i = tid; i < 2 * GROUP * ld; i += 256) hb[i] = 0;
  }
  float c_reg[MAXTW][4], h_reg[MAXTW][4];
  float pre[2][MAXTW][4][4];                           // input-to-hidden pre-activations, prefetched TWO steps ahead
  float st[MAXTW][6][4];                               // this step's stash (i,f,g,o,c,h), stored after the exchange is issued
  int len_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    len_r[r] = b < B ? L.lengths[b] : 0;
  }
#pragma unroll
  for (int j = 0; j < MAXTW; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { c_reg[j][r] = 0.f; h_reg[j][r] = 0.f; }
  const int G4 = 4 * H;

  auto load_pre = [&](float (&dst)[MAXTW][4][4], int step) {
    if (step >= T) return;
    const int t = dir ? T - 1 - step : step;
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
      const int lt = htl + 2 * j, ht = me * TPW + lt;
      const int col = ht * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
        bool act = lt < TPW && ht < nHT && col < H && t < len_r[r];
        int64_t gb = (((int64_t)t * B + b) * 2 + dir) * G4 + col;
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[j][g][r] = act ? D.gates[gb + g * H] : 0.f;
      }
    }
  };
  load_pre(pre[0], 0);
  load_pre(pre[1], 1);
  if (tid == 0) lds_ok = 1;
  __syncthreads();
  __amdgpu_buffer_rsrc_t xr[2];
  xr[0] = make_rsrc(Xb, (unsigned)(GROUP * XW * 2));
  xr[1] = make_rsrc(Xb + (size_t)GROUP * XW * 2, (unsigned)(GROUP * XW * 2));

  // one time step with the pre-activation buffer `P` (static index: the loop below is unrolled by two)
  auto do_step = [&](int step, float (&P)[MAXTW][4][4], int cur) -> bool {
    const int t = dir ? T - 1 - step : step;
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
      const int lt = htl + 2 * j, ht = me * TPW + lt;
      if (lt >= TPW || ht >= nHT) continue;
      const int col = ht * 16 + fr;
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned short* hrow = hb + (cur * GROUP + mt * 16 + fr) * ld + fq * 8;
      const bf16x8* wp = reinterpret_cast<const bf16x8*>(Wl) + (size_t)(lt * 4) * KS * 64 + lane;
#pragma unroll 2
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + ks * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wp[(g * KS + ks) * 64], acc[g], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
        bool act = (b < B) && (col < H) && (t < len_r[r]);
        if (act) {
          float gi = sigmoid_fast(acc[0][r] + P[j][0][r]);
          float gf = sigmoid_fast(acc[1][r] + P[j][1][r]);
          float gg = tanh_fast(acc[2][r] + P[j][2][r]);
          float go = sigmoid_fast(acc[3][r] + P[j][3][r]);
          float cn = gf * c_reg[j][r] + gi * gg;
          float hn = go * tanh_fast(cn);
          c_reg[j][r] = cn; h_reg[j][r] = hn;
          st[j][0][r] = gi; st[j][1][r] = gf; st[j][2][r] = gg; st[j][3][r] = go; st[j][4][r] = cn; st[j][5][r] = hn;
        }
        hb[((cur ^ 1) * GROUP + mt * 16 + fq * 4 + r) * ld + col] = f2bf(h_reg[j][r]);
      }
    }
    __syncthreads();                                     // own h slice complete in LDS
    bool ok = true;
    u32x4 gv[8]; int gdst[8];
    if (NC > 1) {
      __amdgpu_buffer_rsrc_t X = (epoch & 1u) ? xr[1] : xr[0];
      const int cpr = TPW * 2;                           // 16-byte chunks of the own slice per row
      const int c0 = me * TPW * 16;
      for (int i = tid; i < GROUP * cpr; i += 256) {
        int row = i / cpr, col = c0 + (i % cpr) * 8;
        if (col >= Hp) continue;                         // last workgroup: tiles past the padded width do not exist
        u32x4 v = *reinterpret_cast<const u32x4*>(&hb[((cur ^ 1) * GROUP + row) * ld + col]);
        st16_sc1(X, (unsigned)((row * XW + col) * 2), v);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
      __syncthreads();
      if (tid == 0) st_flag(flags + (size_t)me * 64, epoch);
      ok = wait_cluster(flags, abort_w, NC, me, epoch, &lds_ok);
      if (ok) {
        const int cprow = Hp / 8;                        // 16-byte chunks per row over all hidden tiles
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          int i = u * 256 + tid;
          gdst[u] = -1;
          if (i < GROUP * cprow) {
            int row = i / cprow, col = (i % cprow) * 8;
            if (col / (TPW * 16) != me) { gdst[u] = ((cur ^ 1) * GROUP + row) * ld + col; gv[u] = ld16_sc1(X, (unsigned)((row * XW + col) * 2)); }
          }
        }
      }
    }
    // stash + the pre-activations of step+2 go out while the gathered h is in flight
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
      const int lt = htl + 2 * j, ht = me * TPW + lt;
      if (lt >= TPW || ht >= nHT) continue;
      const int col = ht * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
        bool inb = (b < B) && (col < H);
        bool act = inb && (t < len_r[r]);
        if (act) {
          int64_t gb = (((int64_t)t * B + b) * 2 + dir) * G4 + col;
          D.gates[gb] = st[j][0][r]; D.gates[gb + H] = st[j][1][r]; D.gates[gb + 2 * H] = st[j][2][r]; D.gates[gb + 3 * H] = st[j][3][r];
          D.cstash[(((int64_t)t * B + b) * 2 + dir) * H + col] = st[j][4][r];
          D.hseq[((int64_t)t * B + b) * 2 * H + dir * H + col] = st[j][5][r];
        } else if (inb) {
          D.hseq[((int64_t)t * B + b) * 2 * H + dir * H + col] = 0.f;
        }
      }
    }
    load_pre(P, step + 2);
    if (NC > 1 && ok) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (gdst[u] >= 0) *reinterpret_cast<u32x4*>(&hb[gdst[u]]) = gv[u];
    }
    __syncthreads();
    return ok;
  };
  {
    int step = 0, cur = 0;
    for (; step + 1 < T; step += 2) {
      if (!do_step(step, pre[0], cur)) { step = T; break; }
      if (!do_step(step + 1, pre[1], cur ^ 1)) { step = T; break; }
    }
    if (step < T) do_step(step, pre[0], cur);
  }
#pragma unroll
  for (int j = 0; j < MAXTW; ++j) {
    const int lt = htl + 2 * j, ht = me * TPW + lt;
    if (lt >= TPW || ht >= nHT) continue;
    const int col = ht * 16 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
      if (b < B && col < H) D.utt[(int64_t)b * 4 * H + (dir * 2 + D.layer) * H + col] = h_reg[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
template <int MAXTW>
__global__ __launch_bounds__(256, 1) void lstm_bwd_cluster_kernel(CLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Where wh = locate(L);
  const CDesc& D = L.d[wh.di];
  const int H = D.H, Hp = D.Hp, KSB = D.KSB, nHT = D.nHT, TPW = D.TPW, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir, me = wh.me;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int mt = wave & 1, htl = wave >> 1;
  const int XW = 4 * Hp;
  const int ld = XW + 8;
  const int ngt = (B + GROUP - 1) / GROUP;

  uint4* Wl = reinterpret_cast<uint4*>(smem);                                              // TPW*KSB*64 x 16 B
  unsigned short* dg = reinterpret_cast<unsigned short*>(smem + (size_t)TPW * KSB * 1024);         // GROUP x ld
  volatile int& lds_ok = *reinterpret_cast<volatile int*>(smem + (size_t)TPW * KSB * 1024 + (size_t)GROUP * ld * 2);

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * 64;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * GROUP * XW * 2;
  {
    const uint4* src = reinterpret_cast<const uint4*>(D.wpack[dir]);       // backward packing [ht*KSB + ks][lane] x 16 B
    const int per_tile = KSB * 64;
    for (int i = tid; i < TPW * per_tile; i += 256) {
      int lt = i / per_tile, ht = me * TPW + lt;
      Wl[i] = ht < nHT ? src[(size_t)ht * per_tile + (i % per_tile)] : uint4{0, 0, 0, 0};
    }
    for (int i = tid; i < GROUP * ld; i += 256) dg[i] = 0;
  }
  float dh_rec[MAXTW][4], dc[MAXTW][4];
  struct Stash { float g[MAXTW][4][4], c[MAXTW][4], cp[MAXTW][4], dh[MAXTW][4]; };
  Stash sb[2];                                          // forward stash of the coming steps, prefetched TWO steps ahead
  float dgv[MAXTW][4][4];                               // this step's gate gradients, stored after the exchange is issued
  int len_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    len_r[r] = b < B ? L.lengths[b] : 0;
  }
#pragma unroll
  for (int j = 0; j < MAXTW; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { dh_rec[j][r] = 0.f; dc[j][r] = 0.f; }
  const int G4 = 4 * H;

  auto load_stash = [&](Stash& S, int step) {
    if (step >= T) return;
    const int t = dir ? step : T - 1 - step;
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
      const int lt = htl + 2 * j, ht = me * TPW + lt;
      const int col = ht * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
        bool act = lt < TPW && ht < nHT && col < H && t < len_r[r];
        int64_t gb = (((int64_t)t * B + b) * 2 + dir) * G4 + col;
#pragma unroll
        for (int g = 0; g < 4; ++g) S.g[j][g][r] = act ? D.gates[gb + g * H] : 0.f;
        S.c[j][r] = act ? D.cstash[(((int64_t)t * B + b) * 2 + dir) * H + col] : 0.f;
        int tp = dir ? t + 1 : t - 1;
        S.cp[j][r] = (act && tp >= 0 && tp < len_r[r]) ? D.cstash[(((int64_t)tp * B + b) * 2 + dir) * H + col] : 0.f;
        float dh = 0.f;
        if (act) {
          if (D.d_hseq) dh += D.d_hseq[((int64_t)t * B + b) * 2 * H + dir * H + col];
          bool fin = dir ? (t == 0) : (t == len_r[r] - 1);
          if (fin) dh += D.utt[(int64_t)b * 4 * H + (dir * 2 + D.layer) * H + col];
        }
        S.dh[j][r] = dh;
      }
    }
  };
  load_stash(sb[0], 0);
  load_stash(sb[1], 1);
  if (tid == 0) lds_ok = 1;
  __syncthreads();
  __amdgpu_buffer_rsrc_t xr[2];
  xr[0] = make_rsrc(Xb, (unsigned)(GROUP * XW * 2));
  xr[1] = make_rsrc(Xb + (size_t)GROUP * XW * 2, (unsigned)(GROUP * XW * 2));
  constexpr int GB = 20;                                // 16-byte gather chunks per thread and batch (text: 19 -> one batch)

  auto do_step = [&](int step, Stash& S) -> bool {
    const int t = dir ? step : T - 1 - step;
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
    // (1) lane-local gate gradients of the own hidden units
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
      const int lt = htl + 2 * j, ht = me * TPW + lt;
      if (lt >= TPW || ht >= nHT) continue;
      const int col = ht * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
        bool act = (b < B) && (col < H) && (t < len_r[r]);
        float dp[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
          float gi = S.g[j][0][r], gf = S.g[j][1][r], gg = S.g[j][2][r], go = S.g[j][3][r];
          float dh = dh_rec[j][r] + S.dh[j][r];
          float tc = tanh_fast(S.c[j][r]);
          float dct = dc[j][r] + dh * go * (1.f - tc * tc);
          dp[0] = dct * gg * gi * (1.f - gi);
          dp[1] = dct * S.cp[j][r] * gf * (1.f - gf);
          dp[2] = dct * gi * (1.f - gg * gg);
          dp[3] = dh * tc * go * (1.f - go);
          dc[j][r] = dct * gf;
        }
        const int row = mt * 16 + fq * 4 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) { dgv[j][g][r] = dp[g]; dg[row * ld + g * Hp + col] = f2bf(dp[g]); }
      }
    }
    __syncthreads();                                     // own dG slice complete in LDS
    bool ok = true;
    if (NC > 1) {
      __amdgpu_buffer_rsrc_t X = (epoch & 1u) ? xr[1] : xr[0];
      const int cpg = TPW * 2;                           // 16-byte chunks per (row, gate) of the own slice
      const int c0 = me * TPW * 16;
      for (int i = tid; i < GROUP * 4 * cpg; i += 256) {
        int row = i / (4 * cpg), rem = i % (4 * cpg);
        int cu = c0 + (rem % cpg) * 8;
        if (cu >= Hp) continue;                          // last workgroup: tiles past the padded width do not exist
        int col = (rem / cpg) * Hp + cu;
        u32x4 v = *reinterpret_cast<const u32x4*>(&dg[row * ld + col]);
        st16_sc1(X, (unsigned)((row * XW + col) * 2), v);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) st_flag(flags + (size_t)me * 64, epoch);
      ok = wait_cluster(flags, abort_w, NC, me, epoch, &lds_ok);
    }
    // this step's dG (fp32, in place over the stash) and the stash of step+2 go out while the gather is in flight
    auto side_traffic = [&]() {
#pragma unroll
      for (int j = 0; j < MAXTW; ++j) {
        const int lt = htl + 2 * j, ht = me * TPW + lt;
        if (lt >= TPW || ht >= nHT) continue;
        const int col = ht * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
          if (b < B && col < H) {
            int64_t gb = (((int64_t)t * B + b) * 2 + dir) * G4 + col;
#pragma unroll
            for (int g = 0; g < 4; ++g) D.gates[gb + g * H] = dgv[j][g][r];
          }
        }
      }
      load_stash(S, step + 2);
    };
    if (NC > 1 && ok) {
      const int cprow = XW / 8;
      const int total = GROUP * cprow;
      bool side_done = false;
      for (int i0 = 0; i0 < total; i0 += 256 * GB) {
        u32x4 v[GB]; int dst[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          int i = i0 + u * 256 + tid;
          dst[u] = -1;
          if (i < total) {
            int row = i / cprow, col = (i % cprow) * 8;
            if ((col % Hp) / (TPW * 16) != me) { dst[u] = row * ld + col; v[u] = ld16_sc1(X_of(epoch, xr), (unsigned)((row * XW + col) * 2)); }
          }
        }
        if (!side_done) { side_traffic(); side_done = true; }
#pragma unroll
        for (int u = 0; u < GB; ++u)
          if (dst[u] >= 0) *reinterpret_cast<u32x4*>(&dg[dst[u]]) = v[u];
      }
      __syncthreads();
    } else {
      side_traffic();
    }
    // (2) dh_{t-1}[own units] = dG(all gate rows) * W_hh[:, own units] out of LDS
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
      const int lt = htl + 2 * j, ht = me * TPW + lt;
      if (lt >= TPW || ht >= nHT) continue;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned short* arow = dg + (mt * 16 + fr) * ld + fq * 8;
      const bf16x8* wp = reinterpret_cast<const bf16x8*>(Wl) + (size_t)lt * KSB * 64 + lane;
#pragma unroll 4
      for (int ks = 0; ks < KSB; ++ks) {
        bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wp[ks * 64], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh_rec[j][r] = acc[r];
    }
    __syncthreads();                                     // dG tile is overwritten by the next step
    return ok;
  };
  {
    int step = 0;
    for (; step + 1 < T; step += 2) {
      if (!do_step(step, sb[0])) { step = T; break; }
      if (!do_step(step + 1, sb[1])) { step = T; break; }
    }
    if (step < T) do_step(step, sb[0]);
  }
}

struct Plan { int TPW, NC, maxtw; size_t lds_f, lds_b; bool ok; };

Plan plan_for(int H) {
  Plan p{};
  int Hp = round_up(H, 16), Kp = round_up(H, 32), KS = Kp / 32, KSB = 4 * Hp / 32, nHT = Hp / 16;
  const size_t cap = 160 * 1024 - 1024;
  p.ok = false;
  for (int t = 6; t >= 1; --t) {
    if (t > nHT && t > 1) continue;
    size_t lf = (size_t)t * 4 * KS * 1024 + (size_t)2 * GROUP * (Kp + 8) * 2 + 16;
    size_t lb = (size_t)t * KSB * 1024 + (size_t)GROUP * (4 * Hp + 8) * 2 + 16;
    if (lf <= cap && lb <= cap) {
      p.TPW = t; p.NC = ceil_div(nHT, t); p.maxtw = ceil_div(t, 2); p.lds_f = lf; p.lds_b = lb; p.ok = p.NC <= 64;
      break;
    }
  }
  return p;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- host entry (used by lstm.hip)
extern "C" int64_t mmda_lstm_xchg_bytes(int H, int B) {
  if (H <= 0 || H > 512 || B <= 0) return MMDA_EINVAL;
  Plan p = plan_for(H);
  if (!p.ok) return 0;                      // no cluster plan: the streaming kernel is used, no exchange buffer needed
  int ngt = ceil_div(B, GROUP);
  return (int64_t)((xchg_bytes(ngt, p.NC, round_up(H, 16)) + 255) & ~(size_t)255);
}

// returns MMDA_OK and sets *used = 1 when the cluster kernels ran; *used = 0 means "not applicable, use the streaming path"
int mmda_lstm_cluster_launch(int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream, bool bwd,
                             int* used) {
  *used = 0;
  if (n > MAXD) return MMDA_OK;
  Plan plans[MAXD];
  int maxtw = 1;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    if (!descs[i].xchg) return MMDA_OK;
    plans[i] = plan_for(descs[i].H);
    if (!plans[i].ok) return MMDA_OK;
    maxtw = plans[i].maxtw > maxtw ? plans[i].maxtw : maxtw;
    size_t l = bwd ? plans[i].lds_b : plans[i].lds_f;
    lds = l > lds ? l : lds;
  }
  if (maxtw > 3) return MMDA_OK;
  const int ngt = ceil_div(B, GROUP);
  int wg_per_group = 0;
  for (int i = 0; i < n; ++i) wg_per_group += 2 * plans[i].NC;
  if (wg_per_group > MAX_WG_PER_LAUNCH) return MMDA_OK;
  const int groups_per_launch = MAX_WG_PER_LAUNCH / wg_per_group;
  hipStream_t s = (hipStream_t)stream;
  for (int g0 = 0; g0 < ngt; g0 += groups_per_launch) {
    CLaunch L;
    L.n = n; L.B = B; L.T = T; L.g0 = g0; L.ng = (ngt - g0) < groups_per_launch ? (ngt - g0) : groups_per_launch;
    L.lengths = lengths; L.epoch_base = descs[0].epoch_base;
    int wg = 0;
    for (int i = 0; i < MAXD; ++i) {
      const mmda_lstm_desc& d = descs[i < n ? i : 0];
      const Plan& p = plans[i < n ? i : 0];
      CDesc& c = L.d[i];
      c.H = d.H; c.Hp = round_up(d.H, 16); c.Kp = round_up(d.H, 32); c.KS = c.Kp / 32; c.KSB = 4 * c.Hp / 32; c.nHT = c.Hp / 16;
      c.TPW = p.TPW; c.NC = p.NC;
      c.gates = d.gates; c.cstash = d.cstash; c.hseq = d.hseq; c.wpack[0] = d.wpack[0]; c.wpack[1] = d.wpack[1];
      c.utt = d.utt; c.layer = d.layer; c.d_hseq = d.d_hseq; c.xchg = (unsigned char*)d.xchg;
      c.wg_begin = wg;
      if (i < n) wg += 2 * L.ng * p.NC;
    }
    dim3 grid(wg), block(256);
#define LAUNCH_C(MT)                                                                                             \
  do {                                                                                                           \
    auto kfn = bwd ? lstm_bwd_cluster_kernel<MT> : lstm_fwd_cluster_kernel<MT>;                                  \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                            (int)lds) != hipSuccess) { (void)hipGetLastError(); }                                \
    hipLaunchKernelGGL(kfn, grid, block, lds, s, L);                                                             \
  } while (0)
    if (maxtw <= 1) LAUNCH_C(1);
    else if (maxtw <= 2) LAUNCH_C(2);
    else LAUNCH_C(3);
#undef LAUNCH_C
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mmda_set_error(bwd ? "mmda_lstm_bwd(cluster)" : "mmda_lstm_fwd(cluster)", e); return MMDA_ELAUNCH; }
  }
  *used = 1;
  return MMDA_OK;
}
