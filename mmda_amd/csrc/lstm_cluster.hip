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
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

constexpr int GROUP = 32;            // samples per cluster (two 16-row MFMA tiles)
constexpr int MAXD = 4;
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr int MAX_WG_PER_LAUNCH = 240;
constexpr int MAXB = 1024;           // blocks per launch (the four-waves-per-tile kernels share CUs: up to 4 blocks each)
// Measured (MOSEI shapes, 108 tiles per 32-sample group): 216 and 432 workgroups (B = 64, 128) beat the one-wave-per-tile kernels by
// 8 % and 6 % of the step, 864 (B = 256, 3.4 workgroups per CU) lose 3 % -- there the CUs' issue slots, not the hand-off, set the pace
// (sleeping longer between polls changes nothing).
constexpr int MAX_WG_QUAD = 512;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

struct CDesc {
  int H, Hp, Kp, KS, KSB, nHT, TPW, NC;
  float* gates; float* cstash; float* hseq; const void* wpack[2]; const void* wpack_c[2]; float* utt; int layer;
  void* dg16;                        // backward, optional: bf16 copy of the gate gradients (gate-minor wave kernel only)
  int dg16_only;                     // ... and no fp32 store of them
  const float* d_hseq;
  unsigned char* xchg;
  int wg_begin;
  int NCw;                           // wave-autonomous forward: workgroups per cluster = ceil(2 * nHT / waves per block)
};
struct CLaunch {
  CDesc d[MAXD];
  int n, B, T, g0, ng;               // batch groups [g0, g0+ng) of this launch
  const int32_t* lengths;
  unsigned epoch_base;
  unsigned long long* dbg;           // diagnostics only: per-workgroup phase cycle sums (NULL in production)
  int gate_minor;                    // `gates` columns are [dir][unit][gate]: 16-byte accesses (see mmda_lstm_desc)
  int xcd_local;                     // the waves may keep the exchange inside one XCD's L2 after checking their placement (see xcc_announce)
  int wpb;                           // wave-autonomous forward: waves per block (1, 2 or 4)
  int no_stash;                      // forward only (evaluation): gates / cell states are not stashed
  short blk2role[MAXB];              // blockIdx -> linear role (-1: no role, exit at once); roles of one cluster share blockIdx % 8
};

// xchg layout per descriptor: [0,64) abort word | flags: (dir, group, wg) x FLAG_STRIDE B | X: (dir, group, parity) x slot
// Flags: four 64-byte lines per workgroup.  The barrier-synchronised kernels signal on line 0; the wave-autonomous forward
// kernel signals per wave on line (m-tile * 2 + local hidden tile).
constexpr size_t FLAG_STRIDE = 256;
__host__ __device__ inline size_t xchg_flags_off() { return 64; }
__host__ __device__ inline size_t xchg_x_off(int ngroups_total, int NC) { return 64 + (size_t)2 * ngroups_total * NC * FLAG_STRIDE; }
// one exchange slot = one (direction, group, parity): forward uses GROUP x 4Hp bf16 of it, backward NC x GROUP x Hp
__host__ __device__ inline size_t xchg_slot(int NC, int Hp) {
  const size_t barrier_form = (size_t)(NC > 4 ? NC : 4) * GROUP * Hp * 2;
  // backward, wave-autonomous: [consumer tile][m-tile][producer tile (count rounded up to even)][64] x 8 B
  const size_t wave_form = (size_t)(Hp / 16) * 2 * (((Hp / 16) + 1) & ~1) * 512;
  return barrier_form > wave_form ? barrier_form : wave_form;
}
__host__ __device__ inline size_t xchg_bytes(int ngroups_total, int NC, int Hp) {
  return xchg_x_off(ngroups_total, NC) + (size_t)2 * ngroups_total * 2 * xchg_slot(NC, Hp);
}

__device__ __forceinline__ void st_rlx(void* p, u64 v) { __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_rlx(const void* p) { return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(void* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag_plain(void* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned ld_flag(const void* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// 16-byte write-through (sc1) store / L1-bypassing (sc1) load through a buffer descriptor built from wave-uniform values
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16); }
__device__ __forceinline__ void st16_plain(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0); }
__device__ __forceinline__ u32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t X_of(unsigned epoch, const __amdgpu_buffer_rsrc_t (&xr)[2]) {
  return (epoch & 1u) ? xr[1] : xr[0];
}

struct Where { int di, dir, grp, me; };
__device__ __forceinline__ Where locate(const CLaunch& L, int role) {
  Where w;
  w.di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < L.n && role >= L.d[i].wg_begin) w.di = i;
  const CDesc& D = L.d[w.di];
  int local = role - D.wg_begin;
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
      while ((int)(ld_flag(flags + (size_t)lane * FLAG_STRIDE) - epoch) < 0) {
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

// Branch-free global I/O: every per-step load/store goes through a buffer descriptor with a 32-bit byte offset; lanes
// that have nothing to do (padding columns, samples past the batch, t >= len_b) use an offset beyond the descriptor, for
// which the hardware returns 0 / drops the store.  (A load under a lane-dependent branch makes hipcc wait vmcnt(0) per
// element, and 64-bit index arithmetic per access costs ~20 VALU instructions.)
constexpr unsigned OOB = 0xFFFFFF00u;
__device__ __forceinline__ float ldf(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
// (cache policy of the bulk stores -- stash, hseq, gate gradients -- as a build-time knob.  Measured at B=32, step in ms: default
//  write-back 0.722, nt 0.740, sc1 write-through 0.751, sc0 sc1 0.753: the end-of-kernel write-back of what is still dirty costs less
//  than write-through traffic beside the hand-offs)
#ifndef MMDA_STASH_AUX
#define MMDA_STASH_AUX 0
#endif
__device__ __forceinline__ void stf(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, MMDA_STASH_AUX);
}
__device__ __forceinline__ f32x4 ldf4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void stf4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, MMDA_STASH_AUX);
}

// ------------------------------------------------------------------------------------------------ forward
// wave -> (m-tile mt = wave&1 : 16 of the group's 32 samples, local hidden tile lt = wave>>1 < TPW)
template <int KSC, bool GM>
__global__ __launch_bounds__(256, 1) void lstm_fwd_cluster_kernel(CLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the serial chain of this kernel is the critical path of the step: its waves issue ahead of any GEMM waves that share the CU
  __builtin_amdgcn_s_setprio(3);
  const int role = L.blk2role[blockIdx.x];
  if (role < 0) return;
  const Where wh = locate(L, role);
  const CDesc& D = L.d[wh.di];
  const int H = D.H, Hp = D.Hp, Kp = D.Kp, KS = D.KS, nHT = D.nHT, TPW = D.TPW, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir, me = wh.me;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int mt = wave & 1, lt = wave >> 1;
  const int ht = me * TPW + lt;
  const bool tile_ok = lt < TPW && ht < nHT;             // wave-uniform
  const int col = ht * 16 + fr;
  const int ld = Kp + 8;
  const int XW = 4 * Hp;
  const int ngt = (B + GROUP - 1) / GROUP;
  const unsigned G4 = 4u * H;
  constexpr bool gm = GM;                               // layout of `gates` (compile time: both forms in one kernel cost registers)

  uint4* Wl = reinterpret_cast<uint4*>(smem);                                            // TPW*4*KS*64 x 16 B
  unsigned short* hb = reinterpret_cast<unsigned short*>(smem + (size_t)TPW * 4 * KS * 1024);   // 2 x GROUP x ld
  volatile int& lds_ok = *reinterpret_cast<volatile int*>(smem + (size_t)TPW * 4 * KS * 1024 + (size_t)2 * GROUP * ld * 2);

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * FLAG_STRIDE;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * xchg_slot(NC, Hp);

  // resident weights: the forward packing is [(ht*4+g)*KS + ks][lane] x 16 B; this workgroup owns tiles me*TPW..+TPW
  {
    const uint4* src = reinterpret_cast<const uint4*>(D.wpack[dir]);
    const int per_tile = 4 * KS * 64;
    if (!(KSC > 0 && KS == KSC))                          // register-resident weights need no LDS copy
    for (int i = tid; i < TPW * per_tile; i += 256) {
      int l2 = i / per_tile, h2 = me * TPW + l2;
      Wl[i] = h2 < nHT ? src[(size_t)h2 * per_tile + (i % per_tile)] : uint4{0, 0, 0, 0};
    }
    for (int i = tid; i < 2 * GROUP * ld; i += 256) hb[i] = 0;
  }
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(D.gates, (unsigned)T * B * 2u * G4 * 4u);
  // cell-state stash: (T,B,2,H) fp32; with the gate-minor layout it is batch-minor-by-4, (T, ceil(B/4), 2, H, 4): the four samples a
  // lane owns are 16 contiguous bytes (one access in the wave-autonomous kernels instead of four)
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(D.cstash, (unsigned)T * (GM ? (unsigned)((B + 3) & ~3) : (unsigned)B) * 2u * H * 4u);
  const unsigned scc = GM ? (unsigned)((B + 3) >> 2) * 2u * H * 16u : (unsigned)B * 2u * H * 4u;      // its byte stride per time step
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(D.hseq, (unsigned)T * B * 2u * H * 4u);
  const unsigned sg = (unsigned)B * 2u * G4 * 4u, sc = (unsigned)B * 2u * H * 4u;   // byte strides per time step (hseq: sc too)
  unsigned og[4], oc[4], oh[4];
  int len_r[4];
  bool inb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    inb[r] = tile_ok && col < H && b < B;
    const int lv = L.lengths[min(b, B - 1)];
    len_r[r] = inb[r] ? lv : 0;
    og[r] = (((unsigned)b * 2u + dir) * G4 + (gm ? col * 4 : col)) * 4u;
    oc[r] = GM ? ((((((unsigned)b >> 2) * 2u + dir) * H + col) << 2) + ((unsigned)b & 3u)) * 4u : (((unsigned)b * 2u + dir) * H + col) * 4u;
    oh[r] = ((unsigned)b * 2u * H + dir * H + col) * 4u;
  }
  float c_reg[4] = {0.f, 0.f, 0.f, 0.f}, h_reg[4] = {0.f, 0.f, 0.f, 0.f};
  float pre[2][4][4];                                   // input-to-hidden pre-activations, prefetched TWO steps ahead
  // register-resident W_hh fragments of this wave's hidden tile (compile-time KS only; see do_step)
  bf16x8 wreg[4][KSC > 0 ? KSC : 1];
  if (KSC > 0 && KS == KSC) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(D.wpack[dir]);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k2 = 0; k2 < (KSC > 0 ? KSC : 1); ++k2)
        wreg[g][k2] = tile_ok ? src[((size_t)(ht * 4 + g) * KSC + k2) * 64 + lane] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }

  auto load_pre = [&](float (&dst)[4][4], int step) {
    const int t = dir ? T - 1 - step : step;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = step < T && t < len_r[r];
      const unsigned o = og[r] + (unsigned)t * sg;
      if (gm) {
        const f32x4 v = ldf4(rg, act ? o : OOB);
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g][r] = v[g];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g][r] = ldf(rg, act ? o + g * H * 4u : OOB);
      }
    }
  };
  load_pre(pre[0], 0);
  load_pre(pre[1], 1);
  if (tid == 0) lds_ok = 1;
  __syncthreads();
  __amdgpu_buffer_rsrc_t xr[2];
  xr[0] = make_rsrc(Xb, (unsigned)(GROUP * XW * 2));
  xr[1] = make_rsrc(Xb + xchg_slot(NC, Hp), (unsigned)(GROUP * XW * 2));

  // all-gather tables (fixed for the whole sequence): X byte offset (OOB = nothing to fetch) and LDS destination
  unsigned goff[8]; int gdst0[8];
  {
    const int cprow = Hp / 8;                            // 16-byte chunks per row over all hidden tiles
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int i = u * 256 + tid;
      int row = i / cprow, cc = (i % cprow) * 8;
      bool want = i < GROUP * cprow && cc / (TPW * 16) != me;
      gdst0[u] = want ? row * ld + cc : -1;
      goff[u] = want ? (unsigned)((row * XW + cc) * 2) : OOB;
    }
  }
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
#define STAMP(i) do { if (L.dbg && tid == 0) { unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - last; last = now_; } } while (0)
  if (L.dbg && tid == 0) last = __builtin_readcyclecounter();

  auto store_stash = [&](const float (&sv)[4][6], int t) {
    if (!tile_ok) return;                                // wave-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = t < len_r[r];
      const unsigned o = act ? og[r] + (unsigned)t * sg : OOB;
      if (gm) {
        stf4(rg, o, f32x4{sv[r][0], sv[r][1], sv[r][2], sv[r][3]});
      } else {
        stf(rg, o, sv[r][0]); stf(rg, act ? o + H * 4u : OOB, sv[r][1]); stf(rg, act ? o + 2u * H * 4u : OOB, sv[r][2]);
        stf(rg, act ? o + 3u * H * 4u : OOB, sv[r][3]);
      }
      stf(rc, act ? oc[r] + (unsigned)t * scc : OOB, sv[r][4]);
      stf(rh, inb[r] ? oh[r] + (unsigned)t * sc : OOB, sv[r][5]);     // zero at padded positions (pad_packed_sequence)
    }
  };
  // one time step with the pre-activation buffer `P` (static index: the loop below is unrolled by two)
  auto do_step = [&](int step, float (&P)[4][4], int cur) -> bool {
    const int t = dir ? T - 1 - step : step;
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
    float sv[4][6];
    if (tile_ok) {
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned short* hrow = hb + (cur * GROUP + mt * 16 + fr) * ld + fq * 8;
      const bf16x8* wp = reinterpret_cast<const bf16x8*>(Wl) + (size_t)(lt * 4) * KS * 64 + lane;
      if (KSC > 0 && KS == KSC) {
        // compile-time trip count (text: 10): the 40 W_hh fragments of this wave's hidden tile live in REGISTERS for the whole
        // sequence (wreg, 160 VGPRs; the kernel runs one wave per SIMD, so 512 are available).  Per step only the ten h
        // fragments come from LDS; re-reading the weights from LDS cost 200 KB of LDS traffic per step and workgroup, which
        // at 128 B/clk was longer than the MFMAs themselves.
        bf16x8 af[KSC];
#pragma unroll
        for (int k2 = 0; k2 < KSC; ++k2) af[k2] = *reinterpret_cast<const bf16x8*>(hrow + k2 * 32);
        __builtin_amdgcn_sched_barrier(0);           // all ten reads in flight before the first MFMA (else hipcc serialises read -> wait -> 4 MFMAs)
#pragma unroll
        for (int k2 = 0; k2 < KSC; ++k2) {
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k2], wreg[g][k2], acc[g], 0, 0, 0);
        }
      } else {
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 a = *reinterpret_cast<const bf16x8*>(hrow + ks * 32);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wp[(g * KS + ks) * 64], acc[g], 0, 0, 0);
        }
      }
      if (L.dbg) { asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0])); STAMP(7); }
      // lane-local cell update, branch-free (inactive lanes compute on zeros and are masked by the selects / OOB stores)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool act = t < len_r[r];
        const float gi = sigmoid_fast(acc[0][r] + P[0][r]);
        const float gf = sigmoid_fast(acc[1][r] + P[1][r]);
        const float gg = tanh_fast(acc[2][r] + P[2][r]);
        const float go = sigmoid_fast(acc[3][r] + P[3][r]);
        const float cn = gf * c_reg[r] + gi * gg;
        const float hn = go * tanh_fast(cn);
        c_reg[r] = act ? cn : c_reg[r];
        h_reg[r] = act ? hn : h_reg[r];
        hb[((cur ^ 1) * GROUP + mt * 16 + fq * 4 + r) * ld + col] = f2bf(h_reg[r]);
        sv[r][0] = gi; sv[r][1] = gf; sv[r][2] = gg; sv[r][3] = go; sv[r][4] = cn; sv[r][5] = act ? hn : 0.f;
      }
      // stash for the backward pass.  Measured: issued here (the publish drain covers it) beats issuing it after the flag store
      // (the polling wave's flag reads queue behind it) or behind the gather loads (its ~40 VMEM issues then sit on the chain).
      store_stash(sv, t);
    }
    STAMP(0);
    __syncthreads();                                     // own h slice complete in LDS
    STAMP(1);
    bool ok = true;
    if (NC > 1) {
      __amdgpu_buffer_rsrc_t X = (epoch & 1u) ? xr[1] : xr[0];
      const int cpr = TPW * 2;                           // 16-byte chunks of the own slice per row
      const int c0 = me * TPW * 16;
      for (int i = tid; i < GROUP * cpr; i += 256) {
        int row = i / cpr, cc = c0 + (i % cpr) * 8;
        if (cc >= Hp) continue;                          // last workgroup: tiles past the padded width do not exist
        u32x4 v = *reinterpret_cast<const u32x4*>(&hb[((cur ^ 1) * GROUP + row) * ld + cc]);
        if (L.xcd_local) st16_plain(X, (unsigned)((row * XW + cc) * 2), v);
        else st16_sc1(X, (unsigned)((row * XW + cc) * 2), v);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
      STAMP(2);
      __syncthreads();
      if (tid == 0) { if (L.xcd_local) st_flag_plain(flags + (size_t)me * FLAG_STRIDE, epoch); else st_flag(flags + (size_t)me * FLAG_STRIDE, epoch); }
      STAMP(3);
      load_pre(P, step + 2);                             // lands while the cluster is being polled
      ok = wait_cluster(flags, abort_w, NC, me, epoch, &lds_ok);
      STAMP(4);
      if (ok) {
        u32x4 gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) gv[u] = ld16_sc1(X, goff[u]);
        STAMP(5);
        const int nb = (cur ^ 1) * GROUP * ld;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (gdst0[u] >= 0) *reinterpret_cast<u32x4*>(&hb[nb + gdst0[u]]) = gv[u];
      }
    } else {
      load_pre(P, step + 2);
    }
    STAMP(6);
    __syncthreads();
    STAMP(7);
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
  if (L.dbg && tid == 0) {
    for (int i = 0; i < 8; ++i) L.dbg[(size_t)role * 8 + i] = ph[i];
    if (L.xcd_local < 0) L.dbg[(size_t)role * 8 + 7] = ((unsigned long long)blockIdx.x << 32) | (unsigned)__builtin_amdgcn_s_getreg(6164);   // XCC_ID
  }
#undef STAMP
  // final hidden state straight into the utterance layout [h1_fwd, h2_fwd, h1_bwd, h2_bwd] (models.py:203)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    if (inb[r]) D.utt[(int64_t)b * 4 * H + (dir * 2 + D.layer) * H + col] = h_reg[r];
  }
}

// Flag poll of the wave-autonomous kernels: lane tau watches the flag of hidden tile tau.  One read in flight: round 1 kept four in
// flight ~130 cycles apart to cut the quantisation of the wait, but every read in flight queues in the polling CU's memory pipeline in
// front of the loads that fetch the data (measured below).  Returns false on timeout / abort (bounded spin).
// (flag reads in flight per polling wave: B=256, all eight groups in one launch, step 2.375 ms with four, 2.354 with two, 2.338 with one)
#ifndef MMDA_FLAGPOLL
#define MMDA_FLAGPOLL 1
#endif
__device__ __forceinline__ bool poll_tiles(const unsigned char* poll_flag, const unsigned char* abort_w, bool watching, unsigned need) {
  if (MMDA_FLAGPOLL == 1) {
    for (unsigned spins = 0;; ++spins) {
      const unsigned f = ld_flag(poll_flag);
      if (__all(!watching || (int)(f - need) >= 0)) return true;
      if (spins > SPIN_LIMIT || ((spins & 255u) == 0 && spins && ld_flag(abort_w) != 0)) return false;
    }
  }
  if (MMDA_FLAGPOLL == 2) {
    unsigned f0 = ld_flag(poll_flag);
    __builtin_amdgcn_s_sleep(2);
    unsigned f1 = ld_flag(poll_flag);
    for (unsigned spins = 0;; spins += 2) {
      if (__all(!watching || (int)(f0 - need) >= 0)) return true;
      f0 = ld_flag(poll_flag);
      if (__all(!watching || (int)(f1 - need) >= 0)) return true;
      f1 = ld_flag(poll_flag);
      if (spins > SPIN_LIMIT || ((spins & 255u) == 0 && spins && ld_flag(abort_w) != 0)) return false;
    }
  }
  unsigned f0 = ld_flag(poll_flag);
  __builtin_amdgcn_s_sleep(2);
  unsigned f1 = ld_flag(poll_flag);
  __builtin_amdgcn_s_sleep(2);
  unsigned f2 = ld_flag(poll_flag);
  __builtin_amdgcn_s_sleep(2);
  unsigned f3 = ld_flag(poll_flag);
  for (unsigned spins = 0;; spins += 4) {
    if (__all(!watching || (int)(f0 - need) >= 0)) return true;
    f0 = ld_flag(poll_flag);
    if (__all(!watching || (int)(f1 - need) >= 0)) return true;
    f1 = ld_flag(poll_flag);
    if (__all(!watching || (int)(f2 - need) >= 0)) return true;
    f2 = ld_flag(poll_flag);
    if (__all(!watching || (int)(f3 - need) >= 0)) return true;
    f3 = ld_flag(poll_flag);
    if (spins > SPIN_LIMIT || ((spins & 255u) == 0 && spins && ld_flag(abort_w) != 0)) return false;
  }
}

// XCD-local hand-off (wave-autonomous kernels).  A wave exchanges data only with the waves of its own m-tile (19 for text), and
// the host places those on block ids that are equal mod 8, which the dispatcher has been observed -- not promised -- to deal
// to one XCD.  When that holds the exchange can stay inside that XCD's L2: plain stores keep their lines there (sc1 stores write
// through and drop them, so that even a same-XCD reader pays the cross-XCD latency) and the readers' sc1 loads are L2-served.
// Because the placement is not a contract the waves CHECK it: each stores {launch epoch, XCC_ID} next to its flag before its first
// (write-through) publish; at step 1, after the usual poll, every wave reads the ids of all tiles of its m-tile and switches to
// the plain-store form only if they all equal its own.  All waves of the m-tile read the same words, so they agree.
typedef unsigned long long gu64 __attribute__((address_space(1)));
// two floats -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
constexpr unsigned FAR = 0x40000000u;      // added to a valid offset it lands beyond any exchange buffer (and does not wrap)
__device__ __forceinline__ unsigned my_xcc_id() { return __builtin_amdgcn_s_getreg(6164) & 15u; }           // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ void xcc_announce(unsigned char* my_flag, unsigned epoch_base, unsigned xcc) {
  __hip_atomic_store((gu64*)(my_flag + 8), ((unsigned long long)xcc << 32) | epoch_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool xcc_all_local(const unsigned char* tile_flag, bool watching, unsigned epoch_base, unsigned xcc) {
  const unsigned long long v = __hip_atomic_load((const gu64*)(tile_flag + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __all(!watching || v == (((unsigned long long)xcc << 32) | epoch_base));
}

// ------------------------------------------------------------------------------------------------ forward, wave-autonomous form
// Same cluster, same math, no workgroup barrier and no LDS staging.  Every wave owns one (16-sample m-tile, 16-unit hidden
// tile) for the whole sequence and runs on its own:
//   poll    lanes 0..nHT-1 each read the flag of one hidden tile of THIS m-tile (one load instruction covers the cluster)
//   gather  the exchange image is TILE-MAJOR: [hidden tile][m-tile][16 rows][16 units] bf16, 512 contiguous bytes per tile.  The
//           A operand of k-step ks is h[m-tile rows][32ks .. 32ks+31] = tiles 2ks and 2ks+1; in fragment order a lane needs the
//           16 bytes at (row = lane & 15, units 8 (lane >> 4) ..+7): ONE 16-byte sc1 load per lane and k-step straight into the
//           MFMA operand registers -- the gathered h never touches LDS; tiles past the last one read as zero (OOB offset)
//   MFMA    W_hh fragments resident in registers (wreg), counted waits let the MFMAs start as fragments land
//   cell    lane-local, as before
//   publish the wave's 16 x 16 bf16 tile goes through 512 B of wave-private LDS (fragment order -> row order) and out as 32
//           16-byte write-through stores covering whole 64-byte sectors; the wave drains them and raises ITS OWN flag
//           (line m-tile*2 + local tile of its workgroup's flag block)
// NST: compile-time knowledge of "no stash" (evaluation pass): 0 = stash, 1 = none, 2 = decided at run time
template <int KSM, bool GM, int CELL, bool DBG, int NST = 2>
__global__ __launch_bounds__(256, 1) void lstm_fwd_wave_kernel(CLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);
  const int role = L.blk2role[blockIdx.x];
  if (role < 0) return;
  // role -> (descriptor, direction, batch group, first wave-role of this block); wave-role rho = (hidden tile, m-tile)
  Where wh;
  wh.di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < L.n && role >= L.d[i].wg_begin) wh.di = i;
  const CDesc& D = L.d[wh.di];
  {
    const int local = role - D.wg_begin;
    wh.dir = local / (L.ng * D.NCw);
    const int rem = local % (L.ng * D.NCw);
    wh.grp = L.g0 + rem / D.NCw;
    wh.me = rem % D.NCw;
  }
  const int H = D.H, Hp = D.Hp, KS = D.KS, nHT = D.nHT, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __builtin_amdgcn_s_setprio(3);         // a serial chain: win issue arbitration against whatever else shares the CU
  const int fr = lane & 15, fq = lane >> 4;
  const int rho = wh.me * L.wpb + wave;
  const int mt = rho & 1, ht = rho >> 1;
  if (ht >= nHT) return;                                // no tile: nothing to compute, nobody waits for this wave
  const int col = ht * 16 + fr;
  const int ngt = (B + GROUP - 1) / GROUP;
  const unsigned G4 = 4u * H;
  constexpr bool gm = GM;
  unsigned short* Tr = reinterpret_cast<unsigned short*>(smem) + wave * 256;      // 16 x 16 bf16, wave-private

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * FLAG_STRIDE;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * xchg_slot(NC, Hp);
  // one 64-byte flag line per wave-role (the block of NC * FLAG_STRIDE bytes holds >= 2 * nHT lines)
  unsigned char* my_flag = flags + (size_t)rho * 64;
  // lane tau < nHT polls the flag of hidden tile tau for this m-tile
  const int tau = lane < nHT ? lane : 0;
  const unsigned char* poll_flag = flags + (size_t)(tau * 2 + mt) * 64;

  bf16x8 wreg[4][KSM];
  {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(D.wpack[dir]);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k2 = 0; k2 < KSM; ++k2)
        wreg[g][k2] = k2 < KS ? src[((size_t)(ht * 4 + g) * KS + k2) * 64 + lane] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(D.gates, (unsigned)T * B * 2u * G4 * 4u);
  // cell-state stash: (T,B,2,H) fp32; with the gate-minor layout it is batch-minor-by-4, (T, ceil(B/4), 2, H, 4): the four samples a
  // lane owns are 16 contiguous bytes (one access in the wave-autonomous kernels instead of four)
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(D.cstash, (unsigned)T * (GM ? (unsigned)((B + 3) & ~3) : (unsigned)B) * 2u * H * 4u);
  const unsigned scc = GM ? (unsigned)((B + 3) >> 2) * 2u * H * 16u : (unsigned)B * 2u * H * 4u;      // its byte stride per time step
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(D.hseq, (unsigned)T * B * 2u * H * 4u);
  const unsigned sg = (unsigned)B * 2u * G4 * 4u, sc = (unsigned)B * 2u * H * 4u;
  unsigned og[4], oc[4], oh[4];
  int len_r[4];
  bool inb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    inb[r] = col < H && b < B;
    const int lv = L.lengths[min(b, B - 1)];
    len_r[r] = inb[r] ? lv : 0;
    og[r] = (((unsigned)b * 2u + dir) * G4 + (gm ? col * 4 : col)) * 4u;
    oc[r] = GM ? ((((((unsigned)b >> 2) * 2u + dir) * H + col) << 2) + ((unsigned)b & 3u)) * 4u : (((unsigned)b * 2u + dir) * H + col) * 4u;
    oh[r] = ((unsigned)b * 2u * H + dir * H + col) * 4u;
  }
  float c_reg[4] = {0.f, 0.f, 0.f, 0.f}, h_reg[4] = {0.f, 0.f, 0.f, 0.f};
  float pre[2][4][4];
  auto load_pre = [&](float (&dst)[4][4], int step) {
    const int t = dir ? T - 1 - step : step;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = step < T && t < len_r[r];
      const unsigned o = og[r] + (unsigned)t * sg;
      if (gm) {
        const f32x4 v = ldf4(rg, act ? o : OOB);
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g][r] = v[g];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g][r] = ldf(rg, act ? o + g * H * 4u : OOB);
      }
    }
  };
  load_pre(pre[0], 0);
  load_pre(pre[1], 1);
  // ONE descriptor over both parity images (selecting between two descriptors per step makes hipcc keep them in VGPRs and
  // wrap every access in a readfirstlane loop); the parity picks a byte offset instead
  const unsigned slot_b = (unsigned)xchg_slot(NC, Hp);
  const unsigned img_b = (unsigned)nHT * 2u * 512u;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(Xb, slot_b + img_b);
  // fragment of k-step ks: tile 2ks + (fq >> 1), row fr, units 8 (fq & 1) ..+7
  // = frag_base + k2 * 2048 bytes; the second tile of the last k-step may not exist (odd tile count): those lanes read zero
  const unsigned frag_base = (unsigned)(((((fq >> 1) * 2 + mt) * 256) + fr * 16 + 8 * (fq & 1)) * 2);
  const unsigned pub_off = lane < 32 ? (unsigned)((((ht * 2 + mt) * 256) + lane * 8) * 2) : OOB;

  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
#define STAMP(i) do { if (DBG && L.dbg && tid == 0) { unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - last; last = now_; } } while (0)
  if (DBG && L.dbg && tid == 0) last = __builtin_readcyclecounter();

  bool alive = true;
  float sv[4][6];                                       // stash of the previous step, flushed behind the next step's fragment loads
  // off the chain: the stash of step `ps` for the backward pass and the pre-activations of step ps + 2 into the buffer that
  // step used.  Issued right after a step's fragment loads (in front of the poll they would delay every wave's flag reads).
  const bool no_stash = NST == 2 ? L.no_stash != 0 : NST == 1;      // launch-uniform
  auto flush = [&](int ps, float (&Pp)[4][4]) {
    const int t = dir ? T - 1 - ps : ps;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = t < len_r[r];
      const unsigned o = (act && !no_stash) ? og[r] + (unsigned)t * sg : OOB;
      if (!no_stash) {
        if (gm) {
          stf4(rg, o, f32x4{sv[r][0], sv[r][1], sv[r][2], sv[r][3]});
        } else {
          stf(rg, o, sv[r][0]); stf(rg, act ? o + H * 4u : OOB, sv[r][1]); stf(rg, act ? o + 2u * H * 4u : OOB, sv[r][2]);
          stf(rg, act ? o + 3u * H * 4u : OOB, sv[r][3]);
        }
        if (!gm) stf(rc, act ? oc[r] + (unsigned)t * scc : OOB, sv[r][4]);
      }
      stf(rh, inb[r] ? oh[r] + (unsigned)t * sc : OOB, sv[r][5]);     // zero at padded positions (pad_packed_sequence)
    }
    // batch-minor stash: the lane's four samples in one 16-byte store (rows past their length carry the frozen state: never read)
    if (gm && !no_stash) stf4(rc, inb[0] ? oc[0] + (unsigned)t * scc : OOB, f32x4{sv[0][4], sv[1][4], sv[2][4], sv[3][4]});
    load_pre(Pp, ps + 2);
  };
  bool fast = false;                                    // XCD-local hand-off in force (wave-uniform, same in every wave of the m-tile)
  const unsigned xcc = my_xcc_id() ^ ((L.xcd_local == 3 && (ht & 1)) ? 8u : 0u);   // (3: test hook, odd tiles announce a wrong id)
  if (L.xcd_local && lane == 0) xcc_announce(my_flag, L.epoch_base, xcc);
  // One time step.  P: pre-activations of this step; Pp: the buffer the previous step used (refilled for step + 1 by flush).
  // FIRST (step 0: h = 0, nothing to wait for) and FM (hand-off form: 0 = the `fast` variable -- steps 0 and 1, before and while the
  // placement is being checked --, 1 = XCD-local for sure, 2 = write-through for sure) are compile-time tags: the steady-state
  // steps carry no step-number or form branches -- every such branch, taken or not, splits the wave's instruction schedule.
  auto do_step = [&](int step, float (&P)[4][4], float (&Pp)[4][4], auto first_tag, auto fm_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int FM = decltype(fm_tag)::value;
    constexpr bool ALLK = false;          // true: all KSM k-steps unconditionally (measured slower: see the note at the loop)
    const int t = dir ? T - 1 - step : step;
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!FIRST) {
      // every hidden tile of this m-tile must have published the previous step
      const unsigned need = epoch - 1u;
      alive = poll_tiles(poll_flag, abort_w, lane < nHT, need);
      if (!alive) { if (lane == 0) st_flag(abort_w, 1u); return; }
      if (FM == 0 && L.xcd_local && step == 1) fast = xcc_all_local(poll_flag, lane < nHT, L.epoch_base, xcc);
      STAMP(0);
      const unsigned par = (need & 1u) * slot_b;
      bf16x8 af[KSM];
#pragma unroll
      for (int k2 = 0; k2 < KSM; ++k2)
        af[k2] = __builtin_bit_cast(bf16x8, ld16_sc1(xr, (k2 < KS && 2 * k2 + (fq >> 1) < nHT) ? par + frag_base + (unsigned)k2 * 2048u : OOB));
      flush(step - 1, Pp);
      STAMP(1);
      // (skipping the k-steps past a narrow modality's depth by a wave-uniform test per k-step; issuing all of them on zero operands
      // instead was measured slower)
#pragma unroll
      for (int k2 = 0; k2 < KSM; ++k2) {
        if (ALLK || k2 < KS) {                           // workgroup-uniform
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k2], wreg[g][k2], acc[g], 0, 0, 0);
        }
      }
      if (DBG && L.dbg) { asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0])); STAMP(2); }
    }
    // lane-local cell update, branch-free (inactive lanes compute on zeros and are masked by the selects / OOB stores)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = t < len_r[r];
      const float gi = sigmoid_fast(acc[0][r] + P[0][r]);
      const float gf = sigmoid_fast(acc[1][r] + P[1][r]);
      float gg, go, cn, hn;
      if (CELL == MMDA_CELL_GRU) {       // four-slot GRU (see lstm.hip): slots r, z, x-part of n, h-part of n; stash [r, z, n, q], h
        go = acc[3][r] + P[3][r];
        gg = tanh_fast(acc[2][r] + P[2][r] + gi * go);
        hn = (1.f - gf) * gg + gf * h_reg[r];
        cn = hn;
      } else {
        gg = tanh_fast(acc[2][r] + P[2][r]);
        go = sigmoid_fast(acc[3][r] + P[3][r]);
        cn = gf * c_reg[r] + gi * gg;
        hn = go * tanh_fast(cn);
      }
      c_reg[r] = act ? cn : c_reg[r];
      h_reg[r] = act ? hn : h_reg[r];
      Tr[(fq * 4 + r) * 16 + fr] = f2bf(h_reg[r]);
      sv[r][0] = gi; sv[r][1] = gf; sv[r][2] = gg; sv[r][3] = go; sv[r][4] = c_reg[r]; sv[r][5] = act ? hn : 0.f;
    }
    if (step + 1 < T) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the tile is complete in the wave-private LDS block
      STAMP(3);
      const unsigned par = (epoch & 1u) * slot_b;
      const u32x4 v = *reinterpret_cast<const u32x4*>(&Tr[(lane & 31) * 8]);
      const bool fst = FM == 0 ? fast : FM == 1;
      if (fst) st16_plain(xr, pub_off == OOB ? OOB : par + pub_off, v);
      else st16_sc1(xr, pub_off == OOB ? OOB : par + pub_off, v);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's stores have landed (in L2 / written through)
      STAMP(4);
      if (lane == 0) { if (fst) st_flag_plain(my_flag, epoch); else st_flag(my_flag, epoch); }
    }
  };
  {
    typedef std::integral_constant<bool, true> TrueT;
    typedef std::integral_constant<bool, false> FalseT;
    typedef std::integral_constant<int, 0> FmVar;
    typedef std::integral_constant<int, 1> FmLocal;
    typedef std::integral_constant<int, 2> FmThrough;
    // pre[0] holds step 0, pre[1] step 1; afterwards flush() refills the buffer of step s - 1 with step s + 1
    int step = 0;
    if (T > 0) { do_step(0, pre[0], pre[1], TrueT{}, FmVar{}); step = 1; }
    if (T > 1 && alive) { do_step(1, pre[1], pre[0], FalseT{}, FmVar{}); step = 2; }
    if (fast) {
      for (; step + 1 < T && alive; step += 2) {
        do_step(step, pre[0], pre[1], FalseT{}, FmLocal{});
        if (alive) do_step(step + 1, pre[1], pre[0], FalseT{}, FmLocal{});
      }
      if (step < T && alive) { do_step(step, pre[0], pre[1], FalseT{}, FmLocal{}); ++step; }
    } else {
      for (; step + 1 < T && alive; step += 2) {
        do_step(step, pre[0], pre[1], FalseT{}, FmThrough{});
        if (alive) do_step(step + 1, pre[1], pre[0], FalseT{}, FmThrough{});
      }
      if (step < T && alive) { do_step(step, pre[0], pre[1], FalseT{}, FmThrough{}); ++step; }
    }
    if (alive && T > 0) flush(T - 1, (T - 1) & 1 ? pre[1] : pre[0]);
  }
  if (DBG && L.dbg && tid == 0)
    for (int i = 0; i < 8; ++i) L.dbg[(size_t)role * 8 + i] = ph[i];
#undef STAMP
  // final hidden state straight into the utterance layout [h1_fwd, h2_fwd, h1_bwd, h2_bwd] (models.py:203)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    if (inb[r]) D.utt[(int64_t)b * 4 * H + (dir * 2 + D.layer) * H + col] = h_reg[r];
  }
}

// ------------------------------------------------------------------------------------------------ backward
// dh_{t-1} = dG_t W_hh.  The workgroup owns the GATE ROWS of its hidden units (K = TPW*64 rows of W_hh, all Hp columns,
// resident in LDS) and computes a PARTIAL dh over all hidden columns from its own dG slice; the partials are then
// reduce-scattered inside the cluster: each workgroup publishes its (GROUP x Hp) bf16 partial and gathers only the
// 2*TPW*16-byte column slices of its own units from the other NC-1 workgroups (18 KB per step for the text LSTM instead of
// the 70 KB an all-gather of dG would cost), sums them in fp32 and continues with its lane-local cell backward.
constexpr int MAXNT = 16;          // n-tiles per wave (two waves share an m-tile): supports nHT <= 32
constexpr int BPU = 8, BGU = 8;    // publish / gather 16-byte chunks per thread (text: 5 and 5)

template <int NTC, bool GM>
__global__ __launch_bounds__(256, 1) void lstm_bwd_cluster_kernel(CLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the serial chain of this kernel is the critical path of the step: its waves issue ahead of any GEMM waves that share the CU
  __builtin_amdgcn_s_setprio(3);
  const int role = L.blk2role[blockIdx.x];
  if (role < 0) return;
  const Where wh = locate(L, role);
  const CDesc& D = L.d[wh.di];
  const int H = D.H, Hp = D.Hp, nHT = D.nHT, TPW = D.TPW, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir, me = wh.me;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int mt = wave & 1, lt = wave >> 1;
  const int ht = me * TPW + lt;
  const bool tile_ok = lt < TPW && ht < nHT;
  const int col = ht * 16 + fr;
  const int KW = TPW * 64, lda = KW + 8, lds = Hp + 8, OW = TPW * 16;
  const int ngt = (B + GROUP - 1) / GROUP;
  const unsigned G4 = 4u * H;
  constexpr bool gm = GM;                               // layout of `gates` (compile time: both forms in one kernel cost registers)

  // LDS: W slice [lt][nt][ks2] fragments | A (own dG) | staging (partial dh) | gathered partials | flag
  size_t off = 0;
  uint4* Wl = reinterpret_cast<uint4*>(smem);                        off += (size_t)TPW * nHT * 2 * 1024;
  unsigned short* Ab = reinterpret_cast<unsigned short*>(smem + off); off += (size_t)GROUP * lda * 2;
  unsigned short* St = reinterpret_cast<unsigned short*>(smem + off); off += (size_t)GROUP * lds * 2;
  unsigned short* Gb = reinterpret_cast<unsigned short*>(smem + off); off += (size_t)(NC > 1 ? NC - 1 : 1) * GROUP * OW * 2;
  volatile int& lds_ok = *reinterpret_cast<volatile int*>(smem + off);

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * FLAG_STRIDE;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * xchg_slot(NC, Hp);
  {
    // cluster-backward packing: [(ht*nHT + nt)*2 + ks2][lane] x 16 B; the tiles of this workgroup are contiguous blocks
    const uint4* src = reinterpret_cast<const uint4*>(D.wpack_c[dir]);
    const int per_tile = nHT * 2 * 64;
    if (!(NTC > 0 && nHT <= 2 * NTC))                     // register-resident weights need no LDS copy
    for (int i = tid; i < TPW * per_tile; i += 256) {
      int l2 = i / per_tile, h2 = me * TPW + l2;
      Wl[i] = h2 < nHT ? src[(size_t)h2 * per_tile + (i % per_tile)] : uint4{0, 0, 0, 0};
    }
    for (int i = tid; i < GROUP * lda; i += 256) Ab[i] = 0;
    for (int i = tid; i < GROUP * lds; i += 256) St[i] = 0;
    for (int i = tid; i < (NC > 1 ? NC - 1 : 1) * GROUP * OW; i += 256) Gb[i] = 0;
  }
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(D.gates, (unsigned)T * B * 2u * G4 * 4u);
  // cell-state stash: (T,B,2,H) fp32; with the gate-minor layout it is batch-minor-by-4, (T, ceil(B/4), 2, H, 4): the four samples a
  // lane owns are 16 contiguous bytes (one access in the wave-autonomous kernels instead of four)
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(D.cstash, (unsigned)T * (GM ? (unsigned)((B + 3) & ~3) : (unsigned)B) * 2u * H * 4u);
  const unsigned scc = GM ? (unsigned)((B + 3) >> 2) * 2u * H * 16u : (unsigned)B * 2u * H * 4u;      // its byte stride per time step
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(const_cast<float*>(D.d_hseq), D.d_hseq ? (unsigned)T * B * 2u * H * 4u : 0u);
  const __amdgpu_buffer_rsrc_t ru = make_rsrc(D.utt, (unsigned)B * 4u * H * 4u);
  const unsigned sg = (unsigned)B * 2u * G4 * 4u, sc = (unsigned)B * 2u * H * 4u;
  unsigned og[4], oc[4], oh[4];
  int len_r[4];
  bool inb[4];
  float d_fin[4];                                       // gradient of the final hidden state of this (sample, unit)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    inb[r] = tile_ok && col < H && b < B;
    const int lv = L.lengths[min(b, B - 1)];
    len_r[r] = inb[r] ? lv : 0;
    og[r] = (((unsigned)b * 2u + dir) * G4 + (gm ? col * 4 : col)) * 4u;
    oc[r] = GM ? ((((((unsigned)b >> 2) * 2u + dir) * H + col) << 2) + ((unsigned)b & 3u)) * 4u : (((unsigned)b * 2u + dir) * H + col) * 4u;
    oh[r] = ((unsigned)b * 2u * H + dir * H + col) * 4u;
    d_fin[r] = ldf(ru, inb[r] ? ((unsigned)b * 4u * H + (dir * 2 + D.layer) * H + col) * 4u : OOB);
  }
  float dh_rec[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
  // register-resident W_hh fragments (compile-time n-tile count only): this wave's n-tiles nt = (wave>>1) + 2i, four k-steps
  // (the gate rows of the workgroup's two hidden tiles) -> 40 fragments = 160 VGPRs, loaded once for the whole sequence
  bf16x8 wreg[NTC > 0 ? NTC : 1][4];
  const bool use_wreg = NTC > 0 && nHT <= 2 * NTC;      // launch-uniform
  if (use_wreg) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(D.wpack_c[dir]);
#pragma unroll
    for (int i = 0; i < (NTC > 0 ? NTC : 1); ++i) {
      const int nt = (wave >> 1) + 2 * i;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int h2 = me * TPW + (ks >> 1);
        const bool have = nt < nHT && ks < TPW * 2 && h2 < nHT;
        wreg[i][ks] = have ? src[((size_t)(h2 * nHT + nt) * 2 + (ks & 1)) * 64 + lane] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  }
  struct Stash { float g[4][4], c[4], cp[4], dh[4]; };
  Stash sb[2];                                          // forward stash of the coming steps, prefetched TWO steps ahead

  auto load_stash = [&](Stash& S, int step) {
    const int t = dir ? step : T - 1 - step;
    const int tp = dir ? t + 1 : t - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = step < T && t < len_r[r];
      const unsigned o = og[r] + (unsigned)t * sg;
      if (gm) {
        const f32x4 v = ldf4(rg, act ? o : OOB);
#pragma unroll
        for (int g = 0; g < 4; ++g) S.g[g][r] = v[g];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) S.g[g][r] = ldf(rg, act ? o + g * H * 4u : OOB);
      }
      S.c[r] = ldf(rc, act ? oc[r] + (unsigned)t * scc : OOB);
      S.cp[r] = ldf(rc, (act && tp >= 0 && tp < len_r[r]) ? oc[r] + (unsigned)tp * scc : OOB);
      S.dh[r] = ldf(rd, act ? oh[r] + (unsigned)t * sc : OOB);          // zero-record descriptor when d_hseq == NULL
    }
  };
  load_stash(sb[0], 0);
  load_stash(sb[1], 1);
  if (tid == 0) lds_ok = 1;
  __amdgpu_buffer_rsrc_t xr[2];
  xr[0] = make_rsrc(Xb, (unsigned)(NC * GROUP * Hp * 2));
  xr[1] = make_rsrc(Xb + xchg_slot(NC, Hp), (unsigned)(NC * GROUP * Hp * 2));
  // publish table: 16-byte chunks of the staged partial that belong to OTHER workgroups' columns -> X[me][row][col]
  unsigned poff[BPU]; int psrc[BPU];
  // gather table: the own-column chunks of every other workgroup's partial -> Gb[src'][row][.]
  unsigned goff[BGU]; int gdst[BGU];
  {
    const int cprow = Hp / 8, cpo = OW / 8;
    const int own_lo = me * OW, own_hi = own_lo + OW;
#pragma unroll
    for (int u = 0; u < BPU; ++u) {
      int i = u * 256 + tid;
      int row = i / cprow, cc = (i % cprow) * 8;
      bool want = NC > 1 && i < GROUP * cprow && !(cc >= own_lo && cc < own_hi);
      psrc[u] = want ? row * lds + cc : 0;
      poff[u] = want ? (unsigned)(((me * GROUP + row) * Hp + cc) * 2) : OOB;
    }
#pragma unroll
    for (int u = 0; u < BGU; ++u) {
      int i = u * 256 + tid;
      int sp = i / (GROUP * cpo), rem = i % (GROUP * cpo);
      int row = rem / cpo, ch = rem % cpo;
      int src = sp < me ? sp : sp + 1;
      bool want = NC > 1 && sp < NC - 1 && own_lo + ch * 8 < Hp;
      gdst[u] = want ? (sp * GROUP + row) * OW + ch * 8 : -1;
      goff[u] = want ? (unsigned)(((src * GROUP + row) * Hp + own_lo + ch * 8) * 2) : OOB;
    }
  }
  __syncthreads();

  auto do_step = [&](int step, Stash& S) -> bool {
    const int t = dir ? step : T - 1 - step;
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
    // (1) lane-local gate gradients of the own hidden units (branch-free); A operand into LDS, fp32 dG in place over the stash
    float dgv[4][4];
    auto store_dg = [&]() {
      if (!tile_ok) return;                              // wave-uniform
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned o = inb[r] ? og[r] + (unsigned)t * sg : OOB;       // zero at padded positions too
        if (gm) {
          stf4(rg, o, f32x4{dgv[r][0], dgv[r][1], dgv[r][2], dgv[r][3]});
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) stf(rg, inb[r] ? o + g * H * 4u : OOB, dgv[r][g]);
        }
      }
    };
    if (tile_ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool act = t < len_r[r];
        const bool fin = dir ? (t == 0) : (t == len_r[r] - 1);
        const float gi = S.g[0][r], gf = S.g[1][r], gg = S.g[2][r], go = S.g[3][r];
        const float dh = dh_rec[r] + S.dh[r] + (fin ? d_fin[r] : 0.f);
        const float tc = tanh_fast(S.c[r]);
        const float dct = dc[r] + dh * go * (1.f - tc * tc);
        float dp[4];
        dp[0] = act ? dct * gg * gi * (1.f - gi) : 0.f;
        dp[1] = act ? dct * S.cp[r] * gf * (1.f - gf) : 0.f;
        dp[2] = act ? dct * gi * (1.f - gg * gg) : 0.f;
        dp[3] = act ? dh * tc * go * (1.f - go) : 0.f;
        dc[r] = act ? dct * gf : dc[r];
        const int row = mt * 16 + fq * 4 + r;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Ab[row * lda + lt * 64 + g * 16 + fr] = f2bf(dp[g]);
          dgv[r][g] = dp[g];                             // fp32 dG goes to memory after the publish (off the chain)
        }
      }
    }
    __syncthreads();                                     // own dG slice (A operand) complete
    // (2) partial dh over ALL hidden columns: (16 x KW) x (KW x 16) per n-tile, W_hh rows of the own units from LDS
    {
      bf16x8 af[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        af[ks] = *reinterpret_cast<const bf16x8*>(&Ab[(mt * 16 + fr) * lda + (ks < TPW * 2 ? ks : 0) * 32 + fq * 8]);
      const bf16x8* wbase = reinterpret_cast<const bf16x8*>(Wl) + lane;
      if (use_wreg) {
#pragma unroll
        for (int i = 0; i < (NTC > 0 ? NTC : 1); ++i) {
          const int nt = (wave >> 1) + 2 * i;
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], wreg[i][ks], acc, 0, 0, 0);
          if (nt < nHT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) St[(mt * 16 + fq * 4 + r) * lds + nt * 16 + fr] = f2bf(acc[r]);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < MAXNT; ++i) {
          const int nt = (wave >> 1) + 2 * i;
          if (nt >= nHT) break;
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            if (ks < TPW * 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], wbase[(((ks >> 1) * nHT + nt) * 2 + (ks & 1)) * 64], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) St[(mt * 16 + fq * 4 + r) * lds + nt * 16 + fr] = f2bf(acc[r]);
        }
      }
    }
    __syncthreads();                                     // staged partial complete
    bool ok = true;
    if (NC > 1) {
      __amdgpu_buffer_rsrc_t X = (epoch & 1u) ? xr[1] : xr[0];
#pragma unroll
      for (int u = 0; u < BPU; ++u) {
        u32x4 v = *reinterpret_cast<const u32x4*>(&St[psrc[u]]);
        if (L.xcd_local) st16_plain(X, poff[u], v);
        else st16_sc1(X, poff[u], v);                    // out-of-range offsets (own columns / past the end) are dropped
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) { if (L.xcd_local) st_flag_plain(flags + (size_t)me * FLAG_STRIDE, epoch); else st_flag(flags + (size_t)me * FLAG_STRIDE, epoch); }
      store_dg();                                        // off the chain: the drain above only waited for the publish stores
      load_stash(S, step + 2);                           // lands while the cluster is being polled
      ok = wait_cluster(flags, abort_w, NC, me, epoch, &lds_ok);
      if (ok) {
        u32x4 gv[BGU];
#pragma unroll
        for (int u = 0; u < BGU; ++u) gv[u] = ld16_sc1(X, goff[u]);
#pragma unroll
        for (int u = 0; u < BGU; ++u)
          if (gdst[u] >= 0) *reinterpret_cast<u32x4*>(&Gb[gdst[u]]) = gv[u];
      }
      __syncthreads();
    } else {
      store_dg();
      load_stash(S, step + 2);
    }
    // (3) reduce: dh_{t-1} of the own units = own partial + the NC-1 gathered partials (fp32 sum of bf16 partials)
    if (tile_ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + fq * 4 + r;
        float sum = bf2f(St[row * lds + me * OW + lt * 16 + fr]);
        for (int sp = 0; sp < NC - 1; ++sp) sum += bf2f(Gb[(sp * GROUP + row) * OW + lt * 16 + fr]);
        dh_rec[r] = sum;
      }
    }
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

// ------------------------------------------------------------------------------------------------ backward, wave-autonomous form
// dh_{t-1} = dG_t W_hh, reduce-scattered as in the barrier kernel but per WAVE: every wave owns one (m-tile, hidden tile) for
// the whole sequence.  From its own dG (16 samples x 64 gate rows, K = 64: two k-steps, A fragments through 2 KB of
// wave-private LDS) it computes the PARTIAL dh for every hidden tile with W_hh fragments that stay in registers, and
// publishes each 16 x 16 partial in the CONSUMER's accumulator-fragment order (lane: unit lane&15, samples (lane>>4)*4..+3,
// four bf16 = 8 bytes per lane).  The consumer of tile nt reads its nHT partials with one 8-byte sc1 load per lane and
// producer, sums them in fp32 straight in fragment order and runs the lane-local cell backward: no workgroup barrier, no
// LDS staging of gathered data, no transposition on the consumer side.
// HDH / D16: compile-time knowledge of "d_hseq is given" (layer 1) and "dG leaves as bf16 only" (0 / 1; 2 = decided at run time)
// PAIR = 1 (four waves per block, large batches): the two waves of a block that share an m-tile -- hidden tiles 2q and 2q + 1 -- add
// their partial dh tiles in LDS before anything leaves the CU: wave 2q keeps the even consumer tiles, wave 2q + 1 the odd ones, each
// hands the other half of its fp32 accumulators over through 10 KB of LDS (16-byte writes, an epoch word, a bounded spin), adds the
// partner's half and publishes ONE bf16 partial per (consumer tile, producer PAIR).  The exchange image then holds (nHT + 1) / 2
// producers per consumer: half the publish stores and half the gather loads per wave and step.  At B = 256 four waves share a CU's
// address unit and a step is paced by its ~45 vector-memory instructions per wave (4.5 us per step against the forward kernel's
// 2.75 at 26): this takes 19 + 19 of them down to 10 + 10.  (A hidden tile without a partner -- the last of an odd count -- keeps
// and publishes everything itself, as pair nHT / 2.)
template <int NTM, bool GM, int CELL, bool DBG, int HDH = 2, int D16 = 2, int PAIR = 0>
__global__ __launch_bounds__(256, 1) void lstm_bwd_wave_kernel(CLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);
  const int role = L.blk2role[blockIdx.x];
  // PAIR: LDS = [4 x 2 KB wave-private dG tiles | 4 x 10 KB hand-over regions | 4 epoch words]
  volatile unsigned* lflag = reinterpret_cast<volatile unsigned*>(smem + 4 * 2048 + 4 * 10240);
  if (PAIR) {
    if ((threadIdx.x & 63) == 0) lflag[threadIdx.x >> 6] = L.epoch_base;      // (LDS is not cleared between launches)
    __syncthreads();                                     // every wave of the block passes here, also the ones that leave below
  }
  if (role < 0) return;
  Where wh;
  wh.di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < L.n && role >= L.d[i].wg_begin) wh.di = i;
  const CDesc& D = L.d[wh.di];
  {
    const int local = role - D.wg_begin;
    wh.dir = local / (L.ng * D.NCw);
    const int rem = local % (L.ng * D.NCw);
    wh.grp = L.g0 + rem / D.NCw;
    wh.me = rem % D.NCw;
  }
  const int H = D.H, Hp = D.Hp, nHT = D.nHT, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __builtin_amdgcn_s_setprio(3);         // a serial chain: win issue arbitration against whatever else shares the CU
  const int fr = lane & 15, fq = lane >> 4;
  const int rho = wh.me * L.wpb + wave;
  const int mt = rho & 1, ht = rho >> 1;
  if (ht >= nHT) return;
  const int col = ht * 16 + fr;
  const int ngt = (B + GROUP - 1) / GROUP;
  const unsigned G4 = 4u * H;
  constexpr bool gm = GM;
  unsigned short* Tr = reinterpret_cast<unsigned short*>(smem) + wave * 1024;     // 16 x 64 bf16, wave-private
  // producer index of this wave in the exchange image, and the number of producers a consumer gathers from
  const int hp = PAIR ? (ht >> 1) : ht;
  const int nP = PAIR ? (nHT + 1) / 2 : nHT;
  const bool has_partner = PAIR && ((ht ^ 1) < nHT);     // (wpb == 4: the partner is wave ^ 2 of this block)
  f32x4* xmine = reinterpret_cast<f32x4*>(smem + 4 * 2048 + wave * 10240) + lane;              // + slot * 64
  const f32x4* xpart = reinterpret_cast<const f32x4*>(smem + 4 * 2048 + (wave ^ 2) * 10240) + lane;

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * FLAG_STRIDE;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * xchg_slot(NC, Hp);
  unsigned char* my_flag = flags + (size_t)rho * 64;
  const int tau = lane < nHT ? lane : 0;
  const unsigned char* poll_flag = flags + (size_t)(tau * 2 + mt) * 64;

  // W_hh fragments: gate rows of the own hidden tile (two k-steps) x every column tile nt; cluster-backward packing
  bf16x8 wreg[NTM][2];
  {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(D.wpack_c[dir]);
#pragma unroll
    for (int nt = 0; nt < NTM; ++nt)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2)
        wreg[nt][ks2] = nt < nHT ? src[((size_t)(ht * nHT + nt) * 2 + ks2) * 64 + lane] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(D.gates, (unsigned)T * B * 2u * G4 * 4u);
  // cell-state stash: (T,B,2,H) fp32; with the gate-minor layout it is batch-minor-by-4, (T, ceil(B/4), 2, H, 4): the four samples a
  // lane owns are 16 contiguous bytes (one access in the wave-autonomous kernels instead of four)
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(D.cstash, (unsigned)T * (GM ? (unsigned)((B + 3) & ~3) : (unsigned)B) * 2u * H * 4u);
  const unsigned scc = GM ? (unsigned)((B + 3) >> 2) * 2u * H * 16u : (unsigned)B * 2u * H * 4u;      // its byte stride per time step
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(const_cast<float*>(D.d_hseq), D.d_hseq ? (unsigned)T * B * 2u * H * 4u : 0u);
  const __amdgpu_buffer_rsrc_t ru = make_rsrc(D.utt, (unsigned)B * 4u * H * 4u);
  const unsigned sg = (unsigned)B * 2u * G4 * 4u, sc = (unsigned)B * 2u * H * 4u;
  unsigned og[4], oc[4], oh[4];
  int len_r[4];
  bool inb[4];
  float d_fin[4];                                       // gradient of the final hidden state of this (sample, unit)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = wh.grp * GROUP + mt * 16 + fq * 4 + r;
    inb[r] = col < H && b < B;
    const int lv = L.lengths[min(b, B - 1)];
    len_r[r] = inb[r] ? lv : 0;
    og[r] = (((unsigned)b * 2u + dir) * G4 + (gm ? col * 4 : col)) * 4u;
    oc[r] = GM ? ((((((unsigned)b >> 2) * 2u + dir) * H + col) << 2) + ((unsigned)b & 3u)) * 4u : (((unsigned)b * 2u + dir) * H + col) * 4u;
    oh[r] = ((unsigned)b * 2u * H + dir * H + col) * 4u;
    d_fin[r] = ldf(ru, inb[r] ? ((unsigned)b * 4u * H + (dir * 2 + D.layer) * H + col) * 4u : OOB);
  }
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  // Forward stash of the NEXT step as loaded (raw), and of the current step folded into the factors the gate gradients are
  // linear in (dv): the folding (tanh, products, masks) depends on the stash alone, so it is done at the end of the previous
  // step, while the other waves' partial sums are still on their way; what stays between "partials arrived" and "own partial
  // published" is a handful of multiplies.
  struct Raw { float g[4][4], cp[4], dh[4]; };
  struct Dv { float a[6][4], dh[4]; };
  Raw raw;
  Dv dv;
  float c_keep[4] = {0.f, 0.f, 0.f, 0.f};               // cell state of the step being derived (LSTM)
  const bool has_dh = HDH == 2 ? D.d_hseq != nullptr : HDH == 1;
  auto load_raw = [&](int step) {
    const int t = dir ? step : T - 1 - step;
    const int tp = dir ? t + 1 : t - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = step < T && t < len_r[r];
      const unsigned o = og[r] + (unsigned)t * sg;
      if (gm) {
        const f32x4 v = ldf4(rg, act ? o : OOB);
#pragma unroll
        for (int g = 0; g < 4; ++g) raw.g[g][r] = v[g];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) raw.g[g][r] = ldf(rg, act ? o + g * H * 4u : OOB);
      }
      // the state of the step that follows in this walk (= the previous one in time); derive() hands it on as that step's own
      // state, so every state is read once
      if (!gm) raw.cp[r] = ldf(rc, (step < T && tp >= 0 && tp < len_r[r]) ? oc[r] + (unsigned)tp * scc : OOB);
      raw.dh[r] = has_dh ? ldf(rd, act ? oh[r] + (unsigned)t * sc : OOB) : 0.f;      // has_dh is workgroup-uniform
    }
    if (gm) {            // batch-minor stash: the lane's four samples in one 16-byte load; derive() masks the rows past their length
      const f32x4 v = ldf4(rc, (step < T && tp >= 0 && tp < T && inb[0]) ? oc[0] + (unsigned)tp * scc : OOB);
#pragma unroll
      for (int r = 0; r < 4; ++r) raw.cp[r] = v[r];
    }
  };
  // raw (step `step`) + c_keep -> dv.  With dh the total gradient of h_t and dc the carried one:
  //   LSTM: dct = dc + dh a4;  dG = [dct a0, dct a1, dct a2, dh a3];  dc' = dct a5
  //   GRU : dh += dc;          dG = [dh a0 a1, dh a2, dh a0, dh a0 a3];  dc' = dh a5       (stash [r, z, n, q], cp = h_{t-1})
  // Everything is zero at inactive (t >= len) positions: the loads were masked, a0 is masked here.  The carry needs no mask: the
  // inactive steps of a sample come first in this walk (forward direction, carry still zero) or last (reverse direction).
  auto derive = [&](int step) {
    const int t = dir ? step : T - 1 - step;
    const int tpd = dir ? t + 1 : t - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (gm) raw.cp[r] = (tpd >= 0 && tpd < len_r[r]) ? raw.cp[r] : 0.f;     // (the scalar form masked its load instead)
      const bool act = step < T && t < len_r[r];
      const bool fin = dir ? (t == 0) : (t == len_r[r] - 1);
      const float gi = raw.g[0][r], gf = raw.g[1][r], gg = raw.g[2][r], go = raw.g[3][r];
      dv.dh[r] = raw.dh[r] + ((act && fin) ? d_fin[r] : 0.f);
      if (CELL == MMDA_CELL_GRU) {
        dv.a[0][r] = act ? (1.f - gf) * (1.f - gg * gg) : 0.f;
        dv.a[1][r] = go * gi * (1.f - gi);
        dv.a[2][r] = (raw.cp[r] - gg) * gf * (1.f - gf);
        dv.a[3][r] = gi;
        dv.a[4][r] = 0.f;
        dv.a[5][r] = gf;
      } else {
        const float tc = tanh_fast(c_keep[r]);
        dv.a[0][r] = gg * gi * (1.f - gi);
        dv.a[1][r] = raw.cp[r] * gf * (1.f - gf);
        dv.a[2][r] = gi * (1.f - gg * gg);
        dv.a[3][r] = tc * go * (1.f - go);
        dv.a[4][r] = go * (1.f - tc * tc);
        dv.a[5][r] = gf;
        c_keep[r] = raw.cp[r];
      }
    }
  };
  if (CELL == MMDA_CELL_LSTM) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t0 = dir ? 0 : T - 1;
      c_keep[r] = ldf(rc, t0 < len_r[r] ? oc[r] + (unsigned)t0 * scc : OOB);
    }
  }
  load_raw(0);
  derive(0);
  load_raw(1);
  // exchange image per parity: [consumer tile nt][m-tile][producer tile][64 lanes] x 8 B
  const unsigned slot_b = (unsigned)xchg_slot(NC, Hp);
  // One region of RS bytes per (consumer tile, m-tile).  Two arrangements of the producers' 512-byte partials inside it:
  //   write-through form : [producer][lane] x 8 B          -- every 128-byte line is written whole by one wave's store
  //   XCD-local form     : [producer pair][lane][2] x 8 B  -- the consumer fetches two producers per 16-byte load (half the load
  //                        instructions of the gather; the 8-byte halves of a line come from two waves, which L2 merges)
  const unsigned RS = (unsigned)((nHT + 1) & ~1) * 512u;
  const unsigned img_b = (unsigned)(nHT * 2) * RS;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(Xb, slot_b + img_b);
  const unsigned gat_base = (unsigned)(ht * 2 + mt) * RS + (unsigned)lane * 8u;          // + producer * 512
  const unsigned pub_base = (unsigned)mt * RS + (unsigned)(hp * 64 + lane) * 8u;         // + consumer tile nt * pub_stride
  const unsigned gat_base2 = (unsigned)(ht * 2 + mt) * RS + (unsigned)lane * 16u;        // + producer pair * 1024
  const unsigned pub_base2 = (unsigned)mt * RS + (unsigned)(hp >> 1) * 1024u + (unsigned)lane * 16u + (unsigned)(hp & 1) * 8u;
  const unsigned pub_stride = 2u * RS;
  bool paired = false;                                  // arrangement of the image the NEXT gather reads (= what the last publish used)

  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
#define STAMP(i) do { if (DBG && L.dbg && tid == 0) { unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - last; last = now_; } } while (0)
  if (DBG && L.dbg && tid == 0) last = __builtin_readcyclecounter();
  bool alive = true;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  float dgv[4][4];                                      // fp32 dG of the previous step, stored behind the next step's gather loads
  const __amdgpu_buffer_rsrc_t rg16 = make_rsrc(D.dg16, D.dg16 ? (unsigned)T * B * 2u * G4 * 2u : 0u);
  const bool has_dg16 = D16 == 2 ? (gm && D.dg16 != nullptr) : (D16 == 1);       // workgroup-uniform
  const bool f32_dg = D16 == 2 ? !(has_dg16 && D.dg16_only) : (D16 == 0);
  auto flush = [&](int ps) {
    const int t = dir ? ps : T - 1 - ps;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned o = inb[r] ? og[r] + (unsigned)t * sg : OOB;       // zero at padded positions too
      if (gm) {
        if (f32_dg) stf4(rg, o, f32x4{dgv[r][0], dgv[r][1], dgv[r][2], dgv[r][3]});
        if (has_dg16) {                                // the same four values rounded to bf16: 8 bytes at half the byte offset
          const u32x2 pk = {pack_bf16x2(dgv[r][0], dgv[r][1]), pack_bf16x2(dgv[r][2], dgv[r][3])};
          __builtin_amdgcn_raw_buffer_store_b64(pk, rg16, inb[r] ? (og[r] + (unsigned)t * sg) >> 1 : OOB, 0, MMDA_STASH_AUX);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) stf(rg, inb[r] ? o + g * H * 4u : OOB, dgv[r][g]);
      }
    }
  };
  bool fast = false;                                    // XCD-local hand-off in force (see xcc_announce)
  const unsigned xcc = my_xcc_id() ^ ((L.xcd_local == 3 && (ht & 1)) ? 8u : 0u);   // (3: test hook, odd tiles announce a wrong id)
  if (L.xcd_local && lane == 0) xcc_announce(my_flag, L.epoch_base, xcc);
  // One step of the walk.  FIRST (step 0: nothing to gather) and FM (hand-off form: 0 = the `fast` / `paired` variables -- steps 0 and
  // 1, before and while the placement is being checked --, 1 = XCD-local and paired for sure, 2 = write-through and unpaired for sure)
  // are compile-time tags: the steady-state steps carry no step-number or form branches (each one, taken or not, splits the wave's
  // instruction schedule; see the forward kernel).
  // ROLE (PAIR kernels; compile-time: the accumulators a wave keeps are static registers): 0 = even tile of a pair (keeps the even
  // consumer tiles), 1 = odd tile (keeps the odd ones), 2 = no partner (keeps everything; also every wave of a PAIR = 0 kernel)
  auto do_step = [&](int step, auto first_tag, auto fm_tag, auto role_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int FM = decltype(fm_tag)::value;
    constexpr int ROLE = decltype(role_tag)::value;
    constexpr int NGQ = PAIR ? NTM / 4 : NTM / 2;        // 16-byte gather loads (two producers each)
    constexpr int NGP = PAIR ? NTM / 2 : NTM;            // 8-byte gather loads (one producer each)
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
    float dh_rec[4] = {0.f, 0.f, 0.f, 0.f};
    if (!FIRST) {
      const unsigned need = epoch - 1u;
      alive = poll_tiles(poll_flag, abort_w, lane < nHT, need);
      if (!alive) { if (lane == 0) st_flag(abort_w, 1u); return; }
      if (FM == 0 && L.xcd_local && step == 1) fast = xcc_all_local(poll_flag, lane < nHT, L.epoch_base, xcc);
      STAMP(0);
      const unsigned par = (need & 1u) * slot_b;
      if (FM == 0 ? paired : FM == 1) {                  // wave-uniform (compile-time in the steady state)
        u32x4 gq[NGQ];
#pragma unroll
        for (int q = 0; q < NGQ; ++q)
          gq[q] = __builtin_amdgcn_raw_buffer_load_b128(xr, gat_base2 + (2 * q < nP ? par + (unsigned)q * 1024u : FAR), 0, 16);
        flush(step - 1);
        load_raw(step + 1);                              // consumed by derive() at the end of this step
        STAMP(1);
#pragma unroll
        for (int q = 0; q < NGQ; ++q) {                  // same summation order as the unpaired form: producer 2q, then 2q + 1
          const bool two = 2 * q + 1 < nP;               // the upper half of an odd count's last pair was written by nobody
          const unsigned b0 = two ? gq[q][2] : 0u, b1 = two ? gq[q][3] : 0u;
          dh_rec[0] += __builtin_bit_cast(float, gq[q][0] << 16);
          dh_rec[1] += __builtin_bit_cast(float, gq[q][0] & 0xffff0000u);
          dh_rec[2] += __builtin_bit_cast(float, gq[q][1] << 16);
          dh_rec[3] += __builtin_bit_cast(float, gq[q][1] & 0xffff0000u);
          dh_rec[0] += __builtin_bit_cast(float, b0 << 16);
          dh_rec[1] += __builtin_bit_cast(float, b0 & 0xffff0000u);
          dh_rec[2] += __builtin_bit_cast(float, b1 << 16);
          dh_rec[3] += __builtin_bit_cast(float, b1 & 0xffff0000u);
        }
      } else {
        u32x2 gv[NGP];
#pragma unroll
        for (int p = 0; p < NGP; ++p)
          gv[p] = __builtin_amdgcn_raw_buffer_load_b64(xr, gat_base + (p < nP ? par + (unsigned)p * 512u : FAR), 0, 16);
        flush(step - 1);
        load_raw(step + 1);                              // consumed by derive() at the end of this step
        STAMP(1);
#pragma unroll
        for (int p = 0; p < NGP; ++p) {
          dh_rec[0] += __builtin_bit_cast(float, gv[p][0] << 16);
          dh_rec[1] += __builtin_bit_cast(float, gv[p][0] & 0xffff0000u);
          dh_rec[2] += __builtin_bit_cast(float, gv[p][1] << 16);
          dh_rec[3] += __builtin_bit_cast(float, gv[p][1] & 0xffff0000u);
        }
      }
    }
    if (DBG && L.dbg) { asm volatile("s_nop 0" :: "v"(dh_rec[0]), "v"(dh_rec[1]), "v"(dh_rec[2]), "v"(dh_rec[3])); STAMP(2); }
    // lane-local gate gradients of the own hidden units: linear in dh / dc with the factors derive() prepared
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float dh = dh_rec[r] + dv.dh[r];
      float dp[4];
      if (CELL == MMDA_CELL_GRU) {
        dh += dc[r];
        const float dpn = dh * dv.a[0][r];
        dp[0] = dpn * dv.a[1][r];
        dp[1] = dh * dv.a[2][r];
        dp[2] = dpn;
        dp[3] = dpn * dv.a[3][r];
        dc[r] = dh * dv.a[5][r];
      } else {
        const float dct = dc[r] + dh * dv.a[4][r];
        dp[0] = dct * dv.a[0][r];
        dp[1] = dct * dv.a[1][r];
        dp[2] = dct * dv.a[2][r];
        dp[3] = dh * dv.a[3][r];
        dc[r] = dct * dv.a[5][r];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Tr[(fq * 4 + r) * 64 + g * 16 + fr] = f2bf(dp[g]);
        dgv[r][g] = dp[g];
      }
    }
    if (step + 1 < T) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the dG tile is complete in the wave-private LDS block
      STAMP(3);
      // A fragments of the two k-steps (gate pairs): row fr, 8 consecutive k at 32 ks2 + 8 fq
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&Tr[fr * 64 + fq * 8]);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&Tr[fr * 64 + 32 + fq * 8]);
      const unsigned par = (epoch & 1u) * slot_b;
      // All 2 x NTM MFMAs first, into accumulators of their own (left to itself the compiler reuses ONE accumulator and waits out
      // every dependent pair before it converts and stores: ~210 cycles per column tile instead of ~32), then convert + store.
      // Column tiles past nHT multiply zero fragments and store to the out-of-range offset.
      f32x4 accs[NTM];
#pragma unroll
      for (int nt = 0; nt < NTM; ++nt) accs[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wreg[nt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NTM; ++nt) accs[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wreg[nt][1], accs[nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (PAIR && ROLE < 2) {
        // hand the partner the tiles IT keeps (consumer tiles of the other parity), take its half of mine: fp32, through LDS
#pragma unroll
        for (int j = 0; j < NTM / 2; ++j) xmine[j * 64] = accs[2 * j + (1 - ROLE)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) lflag[wave] = epoch;
        bool got = false;
        for (unsigned spins = 0; spins < (SPIN_LIMIT << 2); ++spins) {
          if ((int)(lflag[wave ^ 2] - epoch) >= 0) { got = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (!got) { alive = false; if (lane == 0) st_flag(abort_w, 1u); return; }
#pragma unroll
        for (int j = 0; j < NTM / 2; ++j) accs[2 * j + ROLE] += xpart[j * 64];
      }
      // the tile-dependent part of the offset is wave-uniform (scalar select, one vector add); tiles past nHT go out of range
      const bool fst = FM == 0 ? fast : FM == 1;
      constexpr int NPUB = (PAIR && ROLE < 2) ? NTM / 2 : NTM;       // tiles this wave publishes: its parity's, or all
      if (fst) {
#pragma unroll
        for (int j = 0; j < NPUB; ++j) {
          constexpr int dummy = 0; (void)dummy;
          const int nt = (PAIR && ROLE < 2) ? 2 * j + ROLE : j;
          const f32x4 av = accs[(PAIR && ROLE < 2) ? 2 * j + ROLE : j];
          const u32x2 pk = {pack_bf16x2(av[0], av[1]), pack_bf16x2(av[2], av[3])};
          const unsigned so = nt < nHT ? par + (unsigned)nt * pub_stride : FAR;
          __builtin_amdgcn_raw_buffer_store_b64(pk, xr, pub_base2 + so, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NPUB; ++j) {
          const int nt = (PAIR && ROLE < 2) ? 2 * j + ROLE : j;
          const f32x4 av = accs[(PAIR && ROLE < 2) ? 2 * j + ROLE : j];
          const u32x2 pk = {pack_bf16x2(av[0], av[1]), pack_bf16x2(av[2], av[3])};
          const unsigned so = nt < nHT ? par + (unsigned)nt * pub_stride : FAR;
          __builtin_amdgcn_raw_buffer_store_b64(pk, xr, pub_base + so, 0, 16);
        }
      }
      STAMP(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's stores have landed (in L2 / written through)
      STAMP(5);
      if (lane == 0) { if (fst) st_flag_plain(my_flag, epoch); else st_flag(my_flag, epoch); }
      paired = fst;
      __builtin_amdgcn_sched_barrier(0);                             // keep the folding behind the hand-off
      derive(step + 1);
    }
  };
  {
    int step = 0;
    typedef std::integral_constant<bool, true> TrueT;
    typedef std::integral_constant<bool, false> FalseT;
    typedef std::integral_constant<int, 0> FmVar;
    typedef std::integral_constant<int, 1> FmLocal;
    typedef std::integral_constant<int, 2> FmThrough;
    auto walk = [&](auto role_tag) {
      if (T > 0) { do_step(0, TrueT{}, FmVar{}, role_tag); step = 1; }
      if (T > 1 && alive) { do_step(1, FalseT{}, FmVar{}, role_tag); step = 2; }
      if (fast) { for (; step < T && alive; ++step) do_step(step, FalseT{}, FmLocal{}, role_tag); }
      else { for (; step < T && alive; ++step) do_step(step, FalseT{}, FmThrough{}, role_tag); }
    };
    if constexpr (PAIR != 0) {
      if (!has_partner) walk(std::integral_constant<int, 2>{});      // wave-uniform
      else if (ht & 1) walk(std::integral_constant<int, 1>{});
      else walk(std::integral_constant<int, 0>{});
    } else {
      walk(std::integral_constant<int, 2>{});
    }
    if (alive && T > 0) flush(T - 1);
  }
  if (DBG && L.dbg && tid == 0)
    for (int i = 0; i < 8; ++i) L.dbg[(size_t)role * 8 + i] = ph[i];
#undef STAMP
}

// ------------------------------------------------------------------------------------------------ forward, four waves per tile
// The wave-autonomous kernel above spends its step inside ONE wave: 40 MFMAs (~640 cycles), 40 quarter-rate transcendentals plus the
// cell arithmetic for 16 elements per lane (~1200 cycles), 24 memory instructions.  Here a (16-sample m-tile, 16-unit hidden tile)
// belongs to a WORKGROUP of four waves that split the step's serial work four ways:
//   k-split  wave w multiplies k-steps w, w+4, w+8 of h_{t-1} W_hh^T for all four gates (<= 12 MFMAs); it polls the flags of -- and
//            loads fragments from -- the <= 6 hidden tiles those k-steps cover, nothing else
//   reduce   the four partial 16 x 64 gate tiles meet in LDS (one 16-byte write per accumulator row, ONE workgroup barrier per step,
//            double buffered by step parity), summed in a fixed order (wave 0..3)
//   cell     wave w then owns samples 4w..4w+3 of the tile: ONE element per lane (5 exp + 5 rcp instead of 20 + 20), one 16-byte
//            stash store, one cell-state and one h store per lane
//   publish  the wave's four rows of the tile are 128 contiguous bytes of the exchange image (one line): one 2-byte store per lane,
//            drained, then the wave raises ITS flag (dword w of the tile's flag line, bytes 32..47; the placement announcement sits at
//            bytes 48..55 -- the wave-autonomous kernels use bytes 0..15 of the same lines)
// A publish at step s+2 overwrites the image of step s: it happens after the workgroup barrier of step s+2, i.e. after the four
// waves together have seen every tile's four flags for step s+1, hence after every reader of the step-s image has finished with it.
// A wave whose poll times out raises the abort word and from then on stops waiting (its results are garbage, the host discards the
// launch) but keeps arriving at the barriers, so the grid always drains.
// Steady-state data polling: PSETS copies of a wave's three loads in flight, PGAP x 64 cycles between their first issues (one copy:
// before the first).  Measured at B=32 (step, ms): one copy 0.740 (0 / 256 / 512 cycles of first delay: equal, 1024: 0.763), two copies
// issued back to back 0.738, two spaced by 512 / 1024 cycles 0.92 / 0.90, three 1.03, four 1.15 -- every read in flight sits in the
// consumer CU's memory queue in front of the one that will return the data, so polling harder makes the hand-off slower.
#ifndef MMDA_PSETS
#define MMDA_PSETS 1
#define MMDA_PGAP 0
#endif
// ... and 64 cycles of s_sleep between a stale answer and the next question (B=32 step 0.676 -> 0.668 ms; 128 / 256 cycles: 0.676 / 0.692)
#ifndef MMDA_POLL_SLEEP
#define MMDA_POLL_SLEEP 1
#endif
constexpr int PSETS = MMDA_PSETS, PGAP = MMDA_PGAP;
constexpr int QUAD_LDS = 2 * 4 * 16 * 16 * 4 * 4;      // bytes: parity x source wave x 16 rows x 16 units x 4 gates, fp32

__device__ __forceinline__ bool flags4_reached(u32x4 f, unsigned need) {
  return (int)(f[0] - need) >= 0 && (int)(f[1] - need) >= 0 && (int)(f[2] - need) >= 0 && (int)(f[3] - need) >= 0;
}
// lane-per-tile poll of the four wave flags of a tile (one 16-byte L1-bypassing load); three reads in flight (see poll_tiles)
__device__ __forceinline__ bool poll_tiles4(__amdgpu_buffer_rsrc_t fr4, unsigned off, const unsigned char* abort_w, bool watching, unsigned need) {
  u32x4 f0 = ld16_sc1(fr4, off);
  __builtin_amdgcn_s_sleep(2);
  u32x4 f1 = ld16_sc1(fr4, off);
  __builtin_amdgcn_s_sleep(2);
  u32x4 f2 = ld16_sc1(fr4, off);
  for (unsigned spins = 0;; spins += 3) {
    if (__all(!watching || flags4_reached(f0, need))) return true;
    f0 = ld16_sc1(fr4, off);
    if (__all(!watching || flags4_reached(f1, need))) return true;
    f1 = ld16_sc1(fr4, off);
    if (__all(!watching || flags4_reached(f2, need))) return true;
    f2 = ld16_sc1(fr4, off);
    if (spins > SPIN_LIMIT || ((spins & 255u) == 0 && spins && ld_flag(abort_w) != 0)) return false;
  }
}

template <int CELL, int NST>
__global__ __launch_bounds__(256, 4) void lstm_fwd_quad_kernel(CLaunch L) {   // (<= 128 registers: four workgroups share a CU at large batches)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);
  const int role = L.blk2role[blockIdx.x];
  if (role < 0) return;
  Where wh;
  wh.di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < L.n && role >= L.d[i].wg_begin) wh.di = i;
  const CDesc& D = L.d[wh.di];
  {
    const int local = role - D.wg_begin;
    wh.dir = local / (L.ng * D.NCw);
    const int rem = local % (L.ng * D.NCw);
    wh.grp = L.g0 + rem / D.NCw;
    wh.me = rem % D.NCw;
  }
  const int H = D.H, Hp = D.Hp, KS = D.KS, nHT = D.nHT, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;             // MFMA frame: operand row / accumulator column fr, accumulator rows 4 fq ..+3
  const int rho = wh.me;                                // one workgroup per (hidden tile, m-tile)
  const int mt = rho & 1, ht = rho >> 1;
  const int row = 4 * w + fq, b = wh.grp * GROUP + mt * 16 + row, col = ht * 16 + fr;      // cell frame: the lane's one element
  const int ngt = (B + GROUP - 1) / GROUP;
  const unsigned G4 = 4u * H;
  float* red = reinterpret_cast<float*>(smem);

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * FLAG_STRIDE;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * xchg_slot(NC, Hp);
  unsigned char* my_line = flags + (size_t)rho * 64;
  unsigned char* my_flag = my_line + 32 + 4 * w;
  const __amdgpu_buffer_rsrc_t fr4 = make_rsrc(flags, (unsigned)(2 * nHT) * 64u);
  // steps 1 and 2 (flags): lane tau watches the four wave flags of tile tau
  const bool awatch = lane < nHT;
  const unsigned aoff = awatch ? (unsigned)(lane * 2 + mt) * 64u + 32u : OOB;

  bf16x8 wreg[4][3];
  {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(D.wpack[dir]);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k2 = w + 4 * j;
        wreg[g][j] = k2 < KS ? src[((size_t)(ht * 4 + g) * KS + k2) * 64 + lane] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
  }
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(D.gates, (unsigned)T * B * 2u * G4 * 4u);
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(D.cstash, (unsigned)T * (unsigned)((B + 3) & ~3) * 2u * H * 4u);
  const unsigned scc = (unsigned)((B + 3) >> 2) * 2u * H * 16u;
  const __amdgpu_buffer_rsrc_t rh = make_rsrc(D.hseq, (unsigned)T * B * 2u * H * 4u);
  const unsigned sg = (unsigned)B * 2u * G4 * 4u, sc = (unsigned)B * 2u * H * 4u;
  const bool inb = col < H && b < B;
  const int len = inb ? L.lengths[min(b, B - 1)] : 0;
  const unsigned og = (((unsigned)b * 2u + dir) * G4 + col * 4) * 4u;
  const unsigned oc = ((((((unsigned)b >> 2) * 2u + dir) * H + col) << 2) + ((unsigned)b & 3u)) * 4u;
  const unsigned oh = ((unsigned)b * 2u * H + dir * H + col) * 4u;
  float c_reg = 0.f, h_reg = 0.f;
  f32x4 pre[2];
  auto load_pre = [&](f32x4& dst, int step) {
    const int t = dir ? T - 1 - step : step;
    dst = ldf4(rg, (step < T && t < len) ? og + (unsigned)t * sg : OOB);
  };
  load_pre(pre[0], 0);
  load_pre(pre[1], 1);
  const unsigned slot_b = (unsigned)xchg_slot(NC, Hp);
  const unsigned img_b = (unsigned)nHT * 2u * 512u;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(Xb, slot_b + img_b);
  // A fragment of k-step k2: tile 2 k2 + (fq >> 1), row fr, units 8 (fq & 1) ..+7 (the tile-major image of the wave-autonomous form)
  unsigned foff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k2 = w + 4 * j;
    foff[j] = (k2 < KS && 2 * k2 + (fq >> 1) < nHT) ? (unsigned)(((((fq >> 1) * 2 + mt) * 256) + fr * 16 + 8 * (fq & 1)) * 2) + (unsigned)k2 * 2048u : FAR;
  }
  const unsigned pub_off = (unsigned)((((ht * 2 + mt) * 256) + row * 16 + fr) * 2);
  constexpr unsigned TAGS = 0x40004000u;                // bit 14 of both bf16 halves of a dword
  unsigned chk[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) chk[j] = foff[j] != FAR ? TAGS : 0u;
  // LDS: partial of source wave s, element (row, unit): 4 gates = 16 bytes at ((s * 16 + row) * 16 + unit) * 4 floats
  const unsigned wr_base = (unsigned)(((w * 16 + 4 * fq) * 16 + fr) * 4);              // + rr * 64 floats, + parity * 4096
  const unsigned rd_base = (unsigned)((row * 16 + fr) * 4);                              // + s * 1024 floats, + parity * 4096

  bool dead = false;                                    // a poll of this wave timed out (or the launch was aborted): stop waiting
  float sv[6];
  constexpr bool no_stash = NST == 1;
  auto flush = [&](int ps, f32x4& Pp) {
    const int t = dir ? T - 1 - ps : ps;
    const bool act = t < len;
    if (!no_stash) {
      stf4(rg, act ? og + (unsigned)t * sg : OOB, f32x4{sv[0], sv[1], sv[2], sv[3]});
      stf(rc, inb ? oc + (unsigned)t * scc : OOB, sv[4]);        // rows past their length carry the frozen state (never read)
    }
    stf(rh, inb ? oh + (unsigned)t * sc : OOB, sv[5]);           // zero at padded positions (pad_packed_sequence)
    load_pre(Pp, ps + 2);
  };
  bool fast = false;
  const unsigned xcc = my_xcc_id() ^ ((L.xcd_local == 3 && (ht & 1)) ? 8u : 0u);
  if (L.xcd_local && tid == 0)
    __hip_atomic_store((gu64*)(my_line + 48), ((unsigned long long)xcc << 32) | L.epoch_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  auto do_step = [&](int step, f32x4& P, f32x4& Pp, auto first_tag, auto fm_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int FM = decltype(fm_tag)::value;
    const int t = dir ? T - 1 - step : step;
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
    f32x4 z = P;
    if (!FIRST) {
      const unsigned need = epoch - 1u;
      const unsigned par = (need & 1u) * slot_b;
      const unsigned tm = ((need >> 1) & 1u) ? TAGS : 0u;           // the tag bits the image of epoch `need` carries
      u32x4 fa[3];
      if (FM == 0) {                                     // steps 1 and 2: flags (the image may hold a previous launch's tags)
        if (!dead) {
          const bool ok = poll_tiles4(fr4, aoff, abort_w, awatch, need);
          if (!ok) { dead = true; if (lane == 0) st_flag(abort_w, 1u); }
        }
        if (L.xcd_local && step == 1) {
          const unsigned long long v = __hip_atomic_load((const gu64*)(flags + (size_t)((awatch ? lane : 0) * 2 + mt) * 64 + 48), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
          fast = __all(!awatch || v == (((unsigned long long)xcc << 32) | L.epoch_base));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) fa[j] = ld16_sc1(xr, par + foff[j]);
      } else {                                           // steady state: the fragments themselves say when they are there
        // (PSETS copies of the three loads in flight: see the note at PSETS -- one is best)
        u32x4 fs[PSETS][3];
        // the fragments with the expected tag XORed off (what the MFMAs take); an element that still carries the other tag keeps a tag
        // bit -- `stale` -- and a nonexistent tile reads as zero and stays zero (cl = 0)
        auto untag = [&](u32x4 (&x)[3], const u32x4 (&f)[3]) {
          unsigned t = 0;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const unsigned cl = chk[j] ? tm : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[j][i] = f[j][i] ^ cl; t |= x[j][i]; }
          }
          return (t & TAGS) != 0;
        };
        if (PSETS == 1 && PGAP > 0) __builtin_amdgcn_s_sleep(PGAP);
#pragma unroll
        for (int q = 0; q < PSETS; ++q) {
#pragma unroll
          for (int j = 0; j < 3; ++j) fs[q][j] = ld16_sc1(xr, par + foff[j]);
          if (q + 1 < PSETS) __builtin_amdgcn_s_sleep(PGAP);
        }
        bool got = false;
        for (unsigned spins = 0; !got; spins += PSETS) {
#pragma unroll
          for (int q = 0; q < PSETS; ++q) {
            if (!got) {
              if (!__any(untag(fa, fs[q])) || dead) {
                got = true;
              } else {
                __builtin_amdgcn_s_sleep(MMDA_POLL_SLEEP);          // a short pause before asking again (see PSETS)
#pragma unroll
                for (int j = 0; j < 3; ++j) fs[q][j] = ld16_sc1(xr, par + foff[j]);
              }
            }
          }
          if (!got && (spins > SPIN_LIMIT || ((spins & 255u) == 0 && spins && ld_flag(abort_w) != 0))) {
            dead = true; got = true;
            if (lane == 0) st_flag(abort_w, 1u);
          }
        }
      }
      if (FM == 0) {                                     // (flag steps: tags off here)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const unsigned cl = chk[j] ? tm : 0u;
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[j][i] ^= cl;
        }
      }
      bf16x8 af[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) af[j] = __builtin_bit_cast(bf16x8, fa[j]);
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (w + 4 * j < KS) {                            // wave-uniform
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], wreg[g][j], acc[g], 0, 0, 0);
        }
      }
      float* wr = red + (step & 1) * 4096 + wr_base;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) *reinterpret_cast<f32x4*>(wr + rr * 64) = f32x4{acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr]};
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const float* rd = red + (step & 1) * 4096 + rd_base;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(rd), v1 = *reinterpret_cast<const f32x4*>(rd + 1024);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(rd + 2048), v3 = *reinterpret_cast<const f32x4*>(rd + 3072);
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = (((v0[g] + v1[g]) + v2[g]) + v3[g]) + P[g];
    }
    {
      const bool act = t < len;
      const float gi = sigmoid_fast(z[0]);
      const float gf = sigmoid_fast(z[1]);
      float gg, go, cn, hn;
      if (CELL == MMDA_CELL_GRU) {       // four-slot GRU (see lstm.hip): slots r, z, x-part of n, h-part of n; stash [r, z, n, q], h
        go = z[3];
        gg = tanh_fast(z[2] + gi * go);
        hn = (1.f - gf) * gg + gf * h_reg;
        cn = hn;
      } else {
        gg = tanh_fast(z[2]);
        go = sigmoid_fast(z[3]);
        cn = gf * c_reg + gi * gg;
        hn = go * tanh_fast(cn);
      }
      c_reg = act ? cn : c_reg;
      h_reg = act ? hn : h_reg;
      sv[0] = gi; sv[1] = gf; sv[2] = gg; sv[3] = go; sv[4] = c_reg; sv[5] = act ? hn : 0.f;
    }
    if (step + 1 < T) {
      const unsigned par = (epoch & 1u) * slot_b;
      const bool fst = FM == 0 ? fast : FM == 1;
      // |h| < 1, so bit 14 of its bf16 form is free: it carries bit 1 of the epoch -- the value alternates between the successive
      // uses of an image (epochs e, e + 2, ...) and lets a reader tell, element by element, that what it fetched is this step's
      const short hv = (short)(f2bf(h_reg) | (((epoch >> 1) & 1u) << 14));
      if (fst) __builtin_amdgcn_raw_buffer_store_b16(hv, xr, par + pub_off, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b16(hv, xr, par + pub_off, 0, 16);
      if (FM == 0 && step < 2) {                         // steps 0 and 1 are announced by flag as well (read by steps 1 and 2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's stores have landed (in L2 / written through)
        if (lane == 0) { if (fst) st_flag_plain(my_flag, epoch); else st_flag(my_flag, epoch); }
      }
    }
    // the stash of THIS step and the pre-activations of step + 2 (into the buffer this step used), behind the publish: the wave has
    // nothing to do until its neighbours' h arrives
    flush(step, P);
    (void)Pp;
  };
  {
    typedef std::integral_constant<bool, true> TrueT;
    typedef std::integral_constant<bool, false> FalseT;
    typedef std::integral_constant<int, 0> FmVar;
    typedef std::integral_constant<int, 1> FmLocal;
    typedef std::integral_constant<int, 2> FmThrough;
    int step = 0;
    if (T > 0) { do_step(0, pre[0], pre[1], TrueT{}, FmVar{}); step = 1; }
    if (T > 1) { do_step(1, pre[1], pre[0], FalseT{}, FmVar{}); step = 2; }
    if (T > 2) { do_step(2, pre[0], pre[1], FalseT{}, FmVar{}); step = 3; }
    if (fast) {
      for (; step + 1 < T; step += 2) {
        do_step(step, pre[1], pre[0], FalseT{}, FmLocal{});
        do_step(step + 1, pre[0], pre[1], FalseT{}, FmLocal{});
      }
      if (step < T) { do_step(step, pre[1], pre[0], FalseT{}, FmLocal{}); ++step; }
    } else {
      for (; step + 1 < T; step += 2) {
        do_step(step, pre[1], pre[0], FalseT{}, FmThrough{});
        do_step(step + 1, pre[0], pre[1], FalseT{}, FmThrough{});
      }
      if (step < T) { do_step(step, pre[1], pre[0], FalseT{}, FmThrough{}); ++step; }
    }
  }
  // final hidden state straight into the utterance layout [h1_fwd, h2_fwd, h1_bwd, h2_bwd] (models.py:203)
  if (inb) D.utt[(int64_t)b * 4 * H + (dir * 2 + D.layer) * H + col] = h_reg;
}

// ------------------------------------------------------------------------------------------------ backward, four waves per tile
// The same split for dh_{t-1} = dG_t W_hh (see lstm_bwd_wave_kernel for the reduce-scatter this implements).  The workgroup of a
// (m-tile, hidden tile) owns the gate gradients of its 16 x 16 elements:
//   gather   wave w sums the partial dh of rows 4w..4w+3 over all producers.  The image is the write-through form of the wave kernel
//            ([consumer tile][m-tile][producer][lane] x 8 B, a lane's 8 bytes = rows 4 fq..+3 of unit fr), so those rows are 128
//            contiguous bytes per producer: lane (producer slot pl = lane >> 3, unit pair uc = lane & 7) fetches 16 bytes (two units
//            x four rows) of producers pl, pl + 8, pl + 16, adds them in fp32, and a three-stage reduce-scatter across the eight
//            producer slots (lane bits 5, 4, 3) leaves every lane with the sum of ONE element: unit 2 uc + bit 5, row 2 bit 4 + bit 3
//   cell     one element per lane, factors prepared at the end of the previous step (derive)
//   dG tile  16 x 64 bf16 through LDS (each wave writes its four rows, ONE workgroup barrier per step, double buffered)
//   MFMA     wave w multiplies the tile with the W_hh columns of hidden tiles w, w+4, ... (<= 5 tiles, 10 MFMAs), publishes them
// Hand-off without flags, as in the forward kernel -- except that a partial dh has no spare bit of its own: the dG tile is scaled
// by 2^-32 on its way to the matrix cores (a power of two: exact), so that every published bf16 partial is below 2 in magnitude
// and its bit 14 is free for the epoch tag; the gatherer multiplies the fp32 sum by 2^32.  (Gradients beyond 2^33 -- or NaN -- read
// as a tag that never matches or matches early: the first times out into the abort word, both belong to a diverged run.)
// value of lane (l ^ 32) / (l ^ 16) / (l ^ 8).  v_permlane32_swap(x, x) returns (x[l], x[l + 32]) in the lower and (x[l - 32], x[l]) in the
// upper half of the wave; v_permlane16_swap likewise per pair of 16-lane rows; row_ror:8 rotates a row of 16 by 8.
typedef unsigned u2x __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lane_xor32(float v, bool upper) {
  const u2x r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, upper ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor16(float v, bool odd_row) {
  const u2x r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, odd_row ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
}

template <int CELL, int HDH, int D16>
__global__ __launch_bounds__(256, 4) void lstm_bwd_quad_kernel(CLaunch L) {   // (<= 128 registers: four workgroups share a CU at large batches)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);
  const int role = L.blk2role[blockIdx.x];
  if (role < 0) return;
  Where wh;
  wh.di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < L.n && role >= L.d[i].wg_begin) wh.di = i;
  const CDesc& D = L.d[wh.di];
  {
    const int local = role - D.wg_begin;
    wh.dir = local / (L.ng * D.NCw);
    const int rem = local % (L.ng * D.NCw);
    wh.grp = L.g0 + rem / D.NCw;
    wh.me = rem % D.NCw;
  }
  const int H = D.H, Hp = D.Hp, nHT = D.nHT, NC = D.NC;
  const int B = L.B, T = L.T, dir = wh.dir;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;             // MFMA frame
  const int rho = wh.me;
  const int mt = rho & 1, ht = rho >> 1;
  // cell frame: the element the reduce-scatter leaves in this lane
  const int uc = lane & 7, pl = lane >> 3;
  const bool b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
  const int cu = 2 * uc + (b5 ? 1 : 0), crow = 2 * (b4 ? 1 : 0) + (b3 ? 1 : 0);
  const int row = 4 * w + crow, b = wh.grp * GROUP + mt * 16 + row, col = ht * 16 + cu;
  const int ngt = (B + GROUP - 1) / GROUP;
  const unsigned G4 = 4u * H;
  unsigned short* Tr = reinterpret_cast<unsigned short*>(smem);       // [parity][16 rows][64 gate columns] bf16

  unsigned char* abort_w = D.xchg;
  unsigned char* flags = D.xchg + xchg_flags_off() + ((size_t)(dir * ngt + wh.grp) * NC) * FLAG_STRIDE;
  unsigned char* Xb = D.xchg + xchg_x_off(ngt, NC) + ((size_t)(dir * ngt + wh.grp) * 2) * xchg_slot(NC, Hp);
  unsigned char* my_line = flags + (size_t)rho * 64;
  unsigned char* my_flag = my_line + 32 + 4 * w;
  const __amdgpu_buffer_rsrc_t fr4 = make_rsrc(flags, (unsigned)(2 * nHT) * 64u);
  const bool awatch = lane < nHT;
  const unsigned aoff = awatch ? (unsigned)(lane * 2 + mt) * 64u + 32u : OOB;

  // W_hh fragments: gate rows of the own hidden tile (two k-steps) x the column tiles w, w + 4, ...
  bf16x8 wreg[5][2];
  {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(D.wpack_c[dir]);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int nt = w + 4 * j;
        wreg[j][ks2] = nt < nHT ? src[((size_t)(ht * nHT + nt) * 2 + ks2) * 64 + lane] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
  }
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(D.gates, (unsigned)T * B * 2u * G4 * 4u);
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(D.cstash, (unsigned)T * (unsigned)((B + 3) & ~3) * 2u * H * 4u);
  const unsigned scc = (unsigned)((B + 3) >> 2) * 2u * H * 16u;
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(const_cast<float*>(D.d_hseq), D.d_hseq ? (unsigned)T * B * 2u * H * 4u : 0u);
  const unsigned sg = (unsigned)B * 2u * G4 * 4u, sc = (unsigned)B * 2u * H * 4u;
  const bool inb = col < H && b < B;
  const int len = inb ? L.lengths[min(b, B - 1)] : 0;
  const unsigned og = (((unsigned)b * 2u + dir) * G4 + col * 4) * 4u;
  const unsigned oc = ((((((unsigned)b >> 2) * 2u + dir) * H + col) << 2) + ((unsigned)b & 3u)) * 4u;
  const unsigned oh = ((unsigned)b * 2u * H + dir * H + col) * 4u;
  const float d_fin = inb ? D.utt[(size_t)b * 4u * H + (dir * 2 + D.layer) * H + col] : 0.f;
  float dc = 0.f;
  struct Raw { f32x4 g; float cp, dh; } raw;
  struct Dv { float a[6], dh; } dv;
  float c_keep = 0.f;
  const bool has_dh = HDH == 2 ? D.d_hseq != nullptr : HDH == 1;
  auto load_raw = [&](int step) {
    const int t = dir ? step : T - 1 - step;
    const int tp = dir ? t + 1 : t - 1;
    const bool act = step < T && t < len;
    raw.g = ldf4(rg, act ? og + (unsigned)t * sg : OOB);
    raw.cp = ldf(rc, (step < T && tp >= 0 && tp < len) ? oc + (unsigned)tp * scc : OOB);
    raw.dh = has_dh ? ldf(rd, act ? oh + (unsigned)t * sc : OOB) : 0.f;
  };
  auto derive = [&](int step) {                         // see lstm_bwd_wave_kernel
    const int t = dir ? step : T - 1 - step;
    const bool act = step < T && t < len;
    const bool fin = dir ? (t == 0) : (t == len - 1);
    const float gi = raw.g[0], gf = raw.g[1], gg = raw.g[2], go = raw.g[3];
    dv.dh = raw.dh + ((act && fin) ? d_fin : 0.f);
    if (CELL == MMDA_CELL_GRU) {
      dv.a[0] = act ? (1.f - gf) * (1.f - gg * gg) : 0.f;
      dv.a[1] = go * gi * (1.f - gi);
      dv.a[2] = (raw.cp - gg) * gf * (1.f - gf);
      dv.a[3] = gi;
      dv.a[4] = 0.f;
      dv.a[5] = gf;
    } else {
      const float tc = tanh_fast(c_keep);
      dv.a[0] = gg * gi * (1.f - gi);
      dv.a[1] = raw.cp * gf * (1.f - gf);
      dv.a[2] = gi * (1.f - gg * gg);
      dv.a[3] = tc * go * (1.f - go);
      dv.a[4] = go * (1.f - tc * tc);
      dv.a[5] = gf;
      c_keep = raw.cp;
    }
  };
  if (CELL == MMDA_CELL_LSTM) {
    const int t0 = dir ? 0 : T - 1;
    c_keep = ldf(rc, t0 < len ? oc + (unsigned)t0 * scc : OOB);
  }
  load_raw(0);
  derive(0);
  load_raw(1);
  const unsigned slot_b = (unsigned)xchg_slot(NC, Hp);
  const unsigned RS = (unsigned)((nHT + 1) & ~1) * 512u;
  const unsigned img_b = (unsigned)(nHT * 2) * RS;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(Xb, slot_b + img_b);
  constexpr unsigned TAGS = 0x40004000u;
  unsigned goff[3], chk[3];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int p = pl + 8 * l;
    goff[l] = p < nHT ? (unsigned)(ht * 2 + mt) * RS + (unsigned)p * 512u + (unsigned)w * 128u + (unsigned)uc * 16u : FAR;
    chk[l] = p < nHT ? TAGS : 0u;
  }
  const unsigned pub_base = (unsigned)mt * RS + (unsigned)(ht * 64 + lane) * 8u;         // + consumer tile nt * pub_stride
  const unsigned pub_stride = 2u * RS;

  bool dead = false;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  float dgv[4];
  const __amdgpu_buffer_rsrc_t rg16 = make_rsrc(D.dg16, D.dg16 ? (unsigned)T * B * 2u * G4 * 2u : 0u);
  const bool has_dg16 = D16 == 2 ? D.dg16 != nullptr : true;
  const bool f32_dg = D16 == 2 ? !(has_dg16 && D.dg16_only) : false;
  auto flush = [&](int ps) {
    const int t = dir ? ps : T - 1 - ps;
    const unsigned o = inb ? og + (unsigned)t * sg : OOB;               // zero at padded positions too
    if (f32_dg) stf4(rg, o, f32x4{dgv[0], dgv[1], dgv[2], dgv[3]});
    if (has_dg16) {
      const u32x2 pk = {pack_bf16x2(dgv[0], dgv[1]), pack_bf16x2(dgv[2], dgv[3])};
      __builtin_amdgcn_raw_buffer_store_b64(pk, rg16, inb ? (og + (unsigned)t * sg) >> 1 : OOB, 0, MMDA_STASH_AUX);
    }
  };
  bool fast = false;
  const unsigned xcc = my_xcc_id() ^ ((L.xcd_local == 3 && (ht & 1)) ? 8u : 0u);
  if (L.xcd_local && tid == 0)
    __hip_atomic_store((gu64*)(my_line + 48), ((unsigned long long)xcc << 32) | L.epoch_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float up = 4294967296.f, down = 1.f / 4294967296.f;            // 2^32, 2^-32

  auto do_step = [&](int step, auto first_tag, auto fm_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int FM = decltype(fm_tag)::value;
    const unsigned epoch = L.epoch_base + (unsigned)step + 1u;
    float dh_rec = 0.f;
    if (!FIRST) {
      const unsigned need = epoch - 1u;
      const unsigned par = (need & 1u) * slot_b;
      const unsigned tm = ((need >> 1) & 1u) ? TAGS : 0u;
      u32x4 fa[3];
      if (FM == 0) {
        if (!dead) {
          const bool ok = poll_tiles4(fr4, aoff, abort_w, awatch, need);
          if (!ok) { dead = true; if (lane == 0) st_flag(abort_w, 1u); }
        }
        if (L.xcd_local && step == 1) {
          const unsigned long long v = __hip_atomic_load((const gu64*)(flags + (size_t)((awatch ? lane : 0) * 2 + mt) * 64 + 48), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
          fast = __all(!awatch || v == (((unsigned long long)xcc << 32) | L.epoch_base));
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) fa[l] = ld16_sc1(xr, par + goff[l]);
      } else {
        u32x4 fs[PSETS][3];                              // (see the forward kernel)
        auto untag = [&](u32x4 (&x)[3], const u32x4 (&f)[3]) {
          unsigned t = 0;
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            const unsigned cl = chk[l] ? tm : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[l][i] = f[l][i] ^ cl; t |= x[l][i]; }
          }
          return (t & TAGS) != 0;
        };
        if (PSETS == 1 && PGAP > 0) __builtin_amdgcn_s_sleep(PGAP);
#pragma unroll
        for (int q = 0; q < PSETS; ++q) {
#pragma unroll
          for (int l = 0; l < 3; ++l) fs[q][l] = ld16_sc1(xr, par + goff[l]);
          if (q + 1 < PSETS) __builtin_amdgcn_s_sleep(PGAP);
        }
        bool got = false;
        for (unsigned spins = 0; !got; spins += PSETS) {
#pragma unroll
          for (int q = 0; q < PSETS; ++q) {
            if (!got) {
              if (!__any(untag(fa, fs[q])) || dead) {
                got = true;
              } else {
                __builtin_amdgcn_s_sleep(MMDA_POLL_SLEEP);          // a short pause before asking again (see PSETS)
#pragma unroll
                for (int l = 0; l < 3; ++l) fs[q][l] = ld16_sc1(xr, par + goff[l]);
              }
            }
          }
          if (!got && (spins > SPIN_LIMIT || ((spins & 255u) == 0 && spins && ld_flag(abort_w) != 0))) {
            dead = true; got = true;
            if (lane == 0) st_flag(abort_w, 1u);
          }
        }
      }
      if (FM == 0) {                                     // (flag steps: tags off here)
#pragma unroll
        for (int l = 0; l < 3; ++l) {
          const unsigned cl = chk[l] ? tm : 0u;
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[l][i] ^= cl;
        }
      }
      flush(step - 1);
      load_raw(step + 1);                                // consumed by derive() at the end of this step
      // a[unit bit * 4 + row] of units 2 uc, 2 uc + 1, summed over this lane's producers in the order pl, pl + 8, pl + 16
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned x = fa[l][q];                   // (tags already off)
          a[2 * q] += __builtin_bit_cast(float, x << 16);
          a[2 * q + 1] += __builtin_bit_cast(float, x & 0xffff0000u);
        }
      }
      // reduce-scatter over the producer slots: bit 5 picks the unit, bits 4 and 3 the row.  The partner's value comes by
      // v_permlane32_swap / v_permlane16_swap / a DPP row rotation (VALU, a few cycles each; three dependent ds_bpermute round trips
      // were ~0.15 us of every step: tools/micro/permlane_swap.hip pins the swap semantics)
      float k4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = b5 ? a[4 + i] : a[i], send = b5 ? a[i] : a[4 + i];
        k4[i] = keep + lane_xor32(send, b5);
      }
      float k2[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = b4 ? k4[2 + i] : k4[i], send = b4 ? k4[i] : k4[2 + i];
        k2[i] = keep + lane_xor16(send, b4);
      }
      {
        const float keep = b3 ? k2[1] : k2[0], send = b3 ? k2[0] : k2[1];
        dh_rec = (keep + lane_xor8(send)) * up;
      }
    }
    // gate gradients of the lane's element: linear in dh / dc with the factors derive() prepared
    {
      float dh = dh_rec + dv.dh;
      float dp[4];
      if (CELL == MMDA_CELL_GRU) {
        dh += dc;
        const float dpn = dh * dv.a[0];
        dp[0] = dpn * dv.a[1];
        dp[1] = dh * dv.a[2];
        dp[2] = dpn;
        dp[3] = dpn * dv.a[3];
        dc = dh * dv.a[5];
      } else {
        const float dct = dc + dh * dv.a[4];
        dp[0] = dct * dv.a[0];
        dp[1] = dct * dv.a[1];
        dp[2] = dct * dv.a[2];
        dp[3] = dh * dv.a[3];
        dc = dct * dv.a[5];
      }
      unsigned short* tr = Tr + (step & 1) * 1024 + row * 64 + cu;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        tr[g * 16] = f2bf(dp[g] * down);
        dgv[g] = dp[g];
      }
    }
    if (step + 1 < T) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const unsigned short* tr = Tr + (step & 1) * 1024;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&tr[fr * 64 + fq * 8]);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&tr[fr * 64 + 32 + fq * 8]);
      const unsigned par = (epoch & 1u) * slot_b;
      const unsigned tg = ((epoch >> 1) & 1u) ? TAGS : 0u;
      f32x4 accs[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) accs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wreg[j][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 5; ++j) accs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wreg[j][1], accs[j], 0, 0, 0);
      const bool fst = FM == 0 ? fast : FM == 1;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const u32x2 pk = {pack_bf16x2(accs[j][0], accs[j][1]) | tg, pack_bf16x2(accs[j][2], accs[j][3]) | tg};
        const unsigned so = (w + 4 * j) < nHT ? par + (unsigned)(w + 4 * j) * pub_stride : FAR;
        if (fst) __builtin_amdgcn_raw_buffer_store_b64(pk, xr, pub_base + so, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(pk, xr, pub_base + so, 0, 16);
      }
      if (FM == 0 && step < 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) { if (fst) st_flag_plain(my_flag, epoch); else st_flag(my_flag, epoch); }
      }
      derive(step + 1);
    }
  };
  {
    typedef std::integral_constant<bool, true> TrueT;
    typedef std::integral_constant<bool, false> FalseT;
    typedef std::integral_constant<int, 0> FmVar;
    typedef std::integral_constant<int, 1> FmLocal;
    typedef std::integral_constant<int, 2> FmThrough;
    int step = 0;
    if (T > 0) { do_step(0, TrueT{}, FmVar{}); step = 1; }
    if (T > 1) { do_step(1, FalseT{}, FmVar{}); step = 2; }
    if (T > 2) { do_step(2, FalseT{}, FmVar{}); step = 3; }
    if (fast) { for (; step < T; ++step) do_step(step, FalseT{}, FmLocal{}); }
    else { for (; step < T; ++step) do_step(step, FalseT{}, FmThrough{}); }
    if (T > 0) flush(T - 1);
  }
}

struct Plan { int TPW, NC, maxtw; size_t lds_f, lds_b; bool ok; };

Plan plan_for(int H) {
  Plan p{};
  int Hp = round_up(H, 16), Kp = round_up(H, 32), KS = Kp / 32, nHT = Hp / 16;
  const size_t cap = 160 * 1024 - 1024;
  p.ok = false;
  for (int t = 2; t >= 1; --t) {          // <= 2 hidden tiles per workgroup: every wave owns exactly one (m-tile, hidden tile)
    if (t > nHT && t > 1) continue;
    size_t lf = (size_t)t * 4 * KS * 1024 + (size_t)2 * GROUP * (Kp + 8) * 2 + 16;
    const int nc = ceil_div(nHT, t);
    size_t lb = (size_t)t * nHT * 2 * 1024 + (size_t)GROUP * (t * 64 + 8) * 2 + (size_t)GROUP * (Hp + 8) * 2 +
                (size_t)(nc > 1 ? nc - 1 : 1) * GROUP * t * 16 * 2 + 16;
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

static unsigned long long* g_dbg = nullptr;
// diagnostics only (tools/): device buffer of 8 x u64 per workgroup receiving the forward kernel's phase cycle sums
extern "C" int mmda_debug_set_lstm_stamps(void* device_buffer) { g_dbg = (unsigned long long*)device_buffer; return MMDA_OK; }

namespace {
// The wave-autonomous kernels run when every descriptor's W_hh k-steps fit their register-resident form.  Their blocks hold
// 1, 2 or 4 waves (one (hidden tile, m-tile) each): the fewest waves per block that still fit one launch, because the waves
// of a block share the CU's address unit and the per-step memory instructions are what they queue on.
bool wave_form_ok(int n, const mmda_lstm_desc* descs, bool bwd, int* wpb_out, int ngt = 1) {
  static const int no_wave = getenv("MMDA_LSTM_BARRIER_FWD") ? 1 : 0;        // ablation: the barrier-synchronised forward kernel
  static const int force_wpb = getenv("MMDA_LSTM_WPB") ? atoi(getenv("MMDA_LSTM_WPB")) : 0;
  static const int no_wave_b = getenv("MMDA_LSTM_BARRIER_BWD") ? 1 : 0;      // ablation: the barrier-synchronised backward kernel
  bool ok = bwd ? !no_wave_b : !no_wave;
  for (int i = 0; i < n; ++i) ok = ok && round_up(descs[i].H, 32) / 32 <= 10;
  auto count_wgs = [&](int w) { int t = 0; for (int i = 0; i < n; ++i) t += 2 * ceil_div(2 * (round_up(descs[i].H, 16) / 16), w); return t; };
  int wpb = 4;
  if (ok) {
    // ... of ALL ngt batch groups where possible (B = 256: eight groups, 864 waves = 240 four-wave blocks, one wave per SIMD): the
    // groups are independent chains, and a second launch for the groups that did not fit costs a whole extra walk of the sequence,
    // where more waves per CU cost a fraction of a step (they queue on the CU's address unit, not on each other's hand-offs).
    static const int per_group = getenv("MMDA_LSTM_WPB_PER_GROUP") ? 1 : 0;      // ablation: round 1's choice (fit ONE group)
    const int groups = per_group ? 1 : (ngt < 1 ? 1 : ngt);
    wpb = 1;
    while (wpb < 4 && count_wgs(wpb) * groups > MAX_WG_PER_LAUNCH) wpb *= 2;
    if (force_wpb == 1 || force_wpb == 2 || force_wpb == 4) wpb = force_wpb;
    if (count_wgs(wpb) > MAX_WG_PER_LAUNCH) ok = false;
  }
  if (wpb_out) *wpb_out = ok ? wpb : 4;
  return ok;
}

// the checks that decide whether the resident-weights kernels can run these descriptors
bool cluster_applicable(int n, const mmda_lstm_desc* descs, int B, int T, bool bwd, Plan* plans, size_t* lds_out) {
  if (n > MAXD || n <= 0) return false;
  int maxtw = 1;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    if (!descs[i].xchg) return false;
    if (descs[i].cell != descs[0].cell) return false;
    // the GRU cell exists in the wave-autonomous kernels only (and in the streaming kernels of lstm.hip)
    if (descs[i].cell != MMDA_CELL_LSTM && !wave_form_ok(n, descs, bwd, nullptr)) return false;
    if (bwd && (!descs[i].wpack_c[0] || !descs[i].wpack_c[1])) return false;
    if (descs[i].gate_minor != descs[0].gate_minor) return false;
    plans[i] = plan_for(descs[i].H);
    if (!plans[i].ok) return false;
    maxtw = plans[i].maxtw > maxtw ? plans[i].maxtw : maxtw;
    size_t l = bwd ? plans[i].lds_b : plans[i].lds_f;
    lds = l > lds ? l : lds;
  }
  if (maxtw > 1) return false;
  // buffer descriptors address 32-bit byte offsets: every per-step tensor must stay below 4 GiB
  for (int i = 0; i < n; ++i)
    if ((double)T * B * 2.0 * 4.0 * descs[i].H * 4.0 >= 4.0e9) return false;
  int wg_per_group = 0;
  for (int i = 0; i < n; ++i) wg_per_group += 2 * plans[i].NC;
  if (wg_per_group > MAX_WG_PER_LAUNCH) return false;
  *lds_out = lds;
  return true;
}
}  // namespace

extern "C" int mmda_lstm_resident_applicable(int mode, int n, const mmda_lstm_desc* descs, int B, int T, int backward) {
  if (mode != MMDA_BF16 || !descs || B <= 0 || T <= 0) return 0;
  Plan plans[MAXD];
  size_t lds = 0;
  return cluster_applicable(n, descs, B, T, backward != 0, plans, &lds) ? 1 : 0;
}

extern "C" int mmda_lstm_bwd_emits_dg_bf16(int mode, int n, const mmda_lstm_desc* descs, int B, int T) {
  // the wave-autonomous backward kernel with the gate-minor layout is the one that writes mmda_lstm_desc.dg_bf16
  if (mode != MMDA_BF16 || !descs || n <= 0 || n > MAXD || B <= 0 || T <= 0) return 0;
  Plan plans[MAXD];
  size_t lds = 0;
  if (!cluster_applicable(n, descs, B, T, true, plans, &lds)) return 0;
  if (!wave_form_ok(n, descs, true, nullptr)) return 0;
  for (int i = 0; i < n; ++i)
    if (!descs[i].gate_minor) return 0;
  return 1;
}

// returns MMDA_OK and sets *used = 1 when the cluster kernels ran; *used = 0 means "not applicable, use the streaming path"
int mmda_lstm_cluster_launch(int n, const mmda_lstm_desc* descs, int B, int T, const int32_t* lengths, void* stream, bool bwd,
                             int* used) {
  *used = 0;
  Plan plans[MAXD];
  size_t lds = 0;
  if (!cluster_applicable(n, descs, B, T, bwd, plans, &lds)) return MMDA_OK;
  const int ngt = ceil_div(B, GROUP);
  int wpb = 4;
  const bool fwd_wave = wave_form_ok(n, descs, bwd, &wpb, ngt);   // (named for the forward kernel; selects the wave-autonomous form of either pass)
  const bool gru = descs[0].cell == MMDA_CELL_GRU;            // cluster_applicable() admitted GRU only together with the wave form
  // Four waves per tile (lstm_fwd_quad_kernel): gate-minor layout, every group's tiles in one launch at one workgroup per tile.
  // MMDA_LSTM_NO_QUAD: ablation (the one-wave-per-tile kernels).
  bool quad = false;
  {
    static const int no_quad = getenv("MMDA_LSTM_NO_QUAD") ? 1 : 0;
    int tiles = 0;
    bool okq = fwd_wave && !no_quad && g_dbg == nullptr;
    for (int i = 0; i < n; ++i) {
      okq = okq && descs[i].gate_minor && round_up(descs[i].H, 32) / 32 <= 12 && round_up(descs[i].H, 16) / 16 <= 20;
      tiles += 2 * 2 * (round_up(descs[i].H, 16) / 16);
    }
    // every block of the launch must be resident at once: at most what the occupancy query grants per CU (registers, 32 KB of LDS)
    auto per_cu = [](const void* f) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 256, QUAD_LDS) != hipSuccess) { (void)hipGetLastError(); nb = 1; }
      return nb < 1 ? 1 : (nb > 4 ? 4 : nb);
    };
    static const int occ_f = std::min(per_cu(reinterpret_cast<const void*>(lstm_fwd_quad_kernel<MMDA_CELL_LSTM, 0>)),
                                      per_cu(reinterpret_cast<const void*>(lstm_fwd_quad_kernel<MMDA_CELL_GRU, 0>)));
    static const int occ_b = std::min(per_cu(reinterpret_cast<const void*>(lstm_bwd_quad_kernel<MMDA_CELL_LSTM, 2, 2>)),
                                      per_cu(reinterpret_cast<const void*>(lstm_bwd_quad_kernel<MMDA_CELL_GRU, 2, 2>)));
    static const int quad_cap = getenv("MMDA_LSTM_QUAD_CAP") ? atoi(getenv("MMDA_LSTM_QUAD_CAP")) : MAX_WG_QUAD;
    const int cap = std::min(quad_cap, 240 * (bwd ? occ_b : occ_f));
    quad = okq && tiles * ngt <= cap;
    if (quad) wpb = 1;
  }
  int members[MAXD];                       // workgroups per cluster
  int wg_per_group = 0;
  for (int i = 0; i < n; ++i) {
    members[i] = fwd_wave ? ceil_div(2 * (round_up(descs[i].H, 16) / 16), wpb) : plans[i].NC;
    wg_per_group += 2 * members[i];
  }
  const int groups_per_launch = quad ? ngt : MAX_WG_PER_LAUNCH / wg_per_group;
  hipStream_t s = (hipStream_t)stream;
  for (int g0 = 0; g0 < ngt; g0 += groups_per_launch) {
    CLaunch L;
    L.n = n; L.B = B; L.T = T; L.g0 = g0; L.ng = (ngt - g0) < groups_per_launch ? (ngt - g0) : groups_per_launch;
    L.lengths = lengths; L.epoch_base = descs[0].epoch_base; L.dbg = g_dbg;
    L.gate_minor = descs[0].gate_minor ? 1 : 0;
    L.wpb = wpb;
    L.no_stash = 1;
    for (int i = 0; i < n; ++i) L.no_stash = L.no_stash && descs[i].forward_only;
    int wg = 0;
    for (int i = 0; i < MAXD; ++i) {
      const mmda_lstm_desc& d = descs[i < n ? i : 0];
      const Plan& p = plans[i < n ? i : 0];
      CDesc& c = L.d[i];
      c.H = d.H; c.Hp = round_up(d.H, 16); c.Kp = round_up(d.H, 32); c.KS = c.Kp / 32; c.KSB = 4 * c.Hp / 32; c.nHT = c.Hp / 16;
      c.TPW = p.TPW; c.NC = p.NC; c.NCw = members[i < n ? i : 0];
      c.gates = d.gates; c.cstash = d.cstash; c.hseq = d.hseq; c.wpack[0] = d.wpack[0]; c.wpack[1] = d.wpack[1];
      c.wpack_c[0] = d.wpack_c[0]; c.wpack_c[1] = d.wpack_c[1];
      c.utt = d.utt; c.layer = d.layer; c.d_hseq = d.d_hseq; c.xchg = (unsigned char*)d.xchg;
      c.dg16 = (bwd && fwd_wave && d.gate_minor) ? d.dg_bf16 : nullptr;
      c.dg16_only = (c.dg16 && d.dg_bf16_only) ? 1 : 0;
      c.wg_begin = wg;
      if (i < n) wg += 2 * L.ng * members[i];
    }
    // Placement (speed only): blocks b and b + 8 are dealt to the same XCD, so the members of one cluster get block ids that
    // are equal mod 8; clusters go to the XCD with the fewest members so far.
    static const int use_place = getenv("MMDA_NO_PLACEMENT") ? 0 : 1;
    static const int xcd_env = getenv("MMDA_XCD_LOCAL") ? atoi(getenv("MMDA_XCD_LOCAL")) : 1;      // 0: ablation (always write through)
    L.xcd_local = xcd_env;
    for (int b = 0; b < MAXB; ++b) L.blk2role[b] = -1;
    int grid_blocks = wg;
    {
      int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int per_xcd = quad ? 32 * 4 : 32;             // block slots per XCD
      bool fits = use_place && wg <= 8 * per_xcd;
      // (first role, members).  Wave form with one wave per block: a wave exchanges data only with the waves of its own m-tile
      // (roles first + mt, first + mt + 2, ...), so each m-tile is a cluster of its own (19 blocks for text: fits an XCD's 32 CUs).
      const bool by_mt = fwd_wave && wpb == 1;
      const int stride = by_mt ? 2 : 1;
      std::vector<std::pair<int, int>> clusters;
      for (int i = 0; i < n; ++i)
        for (int c2 = 0; c2 < 2 * L.ng; ++c2) {
          const int first = L.d[i].wg_begin + c2 * members[i];
          if (by_mt) { clusters.push_back({first, members[i] / 2}); clusters.push_back({first + 1, members[i] / 2}); }
          else clusters.push_back({first, members[i]});
        }
      std::stable_sort(clusters.begin(), clusters.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second > b.second; });
      short map[MAXB];
      for (int b = 0; b < MAXB; ++b) map[b] = -1;
      int max_slots = 0;
      for (auto& cl : clusters) {
        int x = 0;
        for (int k = 1; k < 8; ++k) if (used[k] < used[x]) x = k;
        if (used[x] + cl.second > per_xcd) { fits = false; break; }
        for (int j = 0; j < cl.second; ++j) map[(used[x] + j) * 8 + x] = (short)(cl.first + j * stride);
        used[x] += cl.second;
        if (used[x] > max_slots) max_slots = used[x];
      }
      if (fits) { for (int b = 0; b < MAXB; ++b) L.blk2role[b] = map[b]; grid_blocks = 8 * max_slots; }
      else { for (int b = 0; b < wg && b < MAXB; ++b) L.blk2role[b] = (short)b; L.xcd_local = 0; }
      // only the wave kernels verify the placement (per m-tile: every wave reads the XCC ids of all hidden tiles of its m-tile)
      // before they rely on it; the barrier-form kernels' unchecked variant stays an experiment (MMDA_XCD_LOCAL=2)
      if (!fwd_wave && xcd_env != 2) L.xcd_local = 0;
    }
    bool bwd_regs = true;                    // every descriptor's n-tiles fit the register-resident form (<= 10 per wave)
    for (int i = 0; i < n; ++i) bwd_regs = bwd_regs && L.d[i].nHT <= 20;
    // Wave form: the kernel needs 2 KB per wave.  With one wave per block it asks for the CU's whole LDS instead: that keeps every
    // LDS-using workgroup of a concurrent kernel (weight-gradient GEMMs and conversions on the side stream) off the ~110 CUs
    // that host a recurrent wave, and leaves them the other ~145.  (MMDA_LSTM_LDS_KB: ablation.)
    static const int lds_kb = getenv("MMDA_LSTM_LDS_KB") ? atoi(getenv("MMDA_LSTM_LDS_KB")) : 160;
    // Only while the launch leaves a good part of the chip free (<= 160 blocks): every block then needs a CU of its own, and the
    // members of a cluster must all be resident at once -- a launch that wants most of the 256 CUs keeps the small allocation, so
    // that its blocks can share CUs if something else (another process on the GPU) holds some.
    const bool reserve = wpb == 1 && grid_blocks <= 160;
    // backward, four waves per block (large batches): the two waves of an m-tile pre-reduce their partial dh tiles in LDS and publish
    // one partial per producer PAIR (lstm_bwd_wave_kernel<..., PAIR = 1>): 4 x 2 KB + 4 x 10 KB + the epoch words.  MMDA_LSTM_PAIR=0: off.
    static const int pair_on = getenv("MMDA_LSTM_PAIR") ? atoi(getenv("MMDA_LSTM_PAIR")) : 1;
    const bool pair_bwd = pair_on && bwd && fwd_wave && !quad && wpb == 4 && L.gate_minor && !gru && g_dbg == nullptr;
    const size_t lds_launch = fwd_wave ? (reserve ? (size_t)(lds_kb < 32 ? 32 : lds_kb > 160 ? 160 : lds_kb) * 1024
                                                  : (quad ? (size_t)QUAD_LDS : (pair_bwd ? (size_t)(4 * 2048 + 4 * 10240 + 64) : (size_t)4 * 2048))) : lds;
    dim3 grid(grid_blocks), block(quad ? 256 : (fwd_wave ? 64 * wpb : 256));
    // the cycle stamps of tools/diag_lstm_phases.py live in a kernel instance of their own (gate-minor LSTM only): even a never-taken
    // branch per phase costs the production kernels scheduling freedom
    const bool dbgk = g_dbg != nullptr && fwd_wave && !gru && L.gate_minor;
    // backward, production form (gate-minor, dG as bf16 only): instances that know at compile time whether d_hseq exists (layer 1: 2,
    // layer 2: 1); anything else takes the instance that decides at run time (0)
    int spec = 0;
    if (bwd && fwd_wave && L.gate_minor) {
      bool all16 = true, any_dh = false, all_dh = true;
      for (int i = 0; i < n; ++i) {
        all16 = all16 && L.d[i].dg16 != nullptr && L.d[i].dg16_only;
        any_dh = any_dh || L.d[i].d_hseq != nullptr;
        all_dh = all_dh && L.d[i].d_hseq != nullptr;
      }
      static const int no_spec = getenv("MMDA_LSTM_NO_SPEC") ? 1 : 0;
      if (all16 && !no_spec) spec = !any_dh ? 1 : (all_dh ? 2 : 0);
    }
#define LAUNCH_C()                                                                                               \
  do {                                                                                                           \
    auto kfn = (quad && bwd) ? (gru ? lstm_bwd_quad_kernel<MMDA_CELL_GRU, 2, 2>                                                                       \
                                    : spec == 1 ? lstm_bwd_quad_kernel<MMDA_CELL_LSTM, 0, 1>                                                           \
                                    : spec == 2 ? lstm_bwd_quad_kernel<MMDA_CELL_LSTM, 1, 1> : lstm_bwd_quad_kernel<MMDA_CELL_LSTM, 2, 2>)              \
             : quad ? (gru ? (L.no_stash ? lstm_fwd_quad_kernel<MMDA_CELL_GRU, 1> : lstm_fwd_quad_kernel<MMDA_CELL_GRU, 0>)                           \
                           : (L.no_stash ? lstm_fwd_quad_kernel<MMDA_CELL_LSTM, 1> : lstm_fwd_quad_kernel<MMDA_CELL_LSTM, 0>))                        \
             : dbgk ? (bwd ? lstm_bwd_wave_kernel<20, true, MMDA_CELL_LSTM, true> : lstm_fwd_wave_kernel<10, true, MMDA_CELL_LSTM, true>)          \
             : gru ? (bwd ? (L.gate_minor ? lstm_bwd_wave_kernel<20, true, MMDA_CELL_GRU, false> : lstm_bwd_wave_kernel<20, false, MMDA_CELL_GRU, false>) \
                          : (L.gate_minor ? lstm_fwd_wave_kernel<10, true, MMDA_CELL_GRU, false> : lstm_fwd_wave_kernel<10, false, MMDA_CELL_GRU, false>)) \
             : bwd ? (fwd_wave ? (L.gate_minor ? (spec == 1 ? (pair_bwd ? lstm_bwd_wave_kernel<20, true, MMDA_CELL_LSTM, false, 0, 1, 1>          \
                                                                        : lstm_bwd_wave_kernel<20, true, MMDA_CELL_LSTM, false, 0, 1>)            \
                                                 : spec == 2 ? (pair_bwd ? lstm_bwd_wave_kernel<20, true, MMDA_CELL_LSTM, false, 1, 1, 1>         \
                                                                        : lstm_bwd_wave_kernel<20, true, MMDA_CELL_LSTM, false, 1, 1>)            \
                                                             : lstm_bwd_wave_kernel<20, true, MMDA_CELL_LSTM, false>)                             \
                                               : lstm_bwd_wave_kernel<20, false, MMDA_CELL_LSTM, false>)                                           \
                    : bwd_regs ? (L.gate_minor ? lstm_bwd_cluster_kernel<10, true> : lstm_bwd_cluster_kernel<10, false>)                \
                              : (L.gate_minor ? lstm_bwd_cluster_kernel<0, true> : lstm_bwd_cluster_kernel<0, false>))                  \
                   : fwd_wave ? (L.gate_minor ? (L.no_stash ? lstm_fwd_wave_kernel<10, true, MMDA_CELL_LSTM, false, 1>                         \
                                                            : lstm_fwd_wave_kernel<10, true, MMDA_CELL_LSTM, false, 0>)                         \
                                              : lstm_fwd_wave_kernel<10, false, MMDA_CELL_LSTM, false>)                                           \
                              : (L.gate_minor ? lstm_fwd_cluster_kernel<10, true> : lstm_fwd_cluster_kernel<10, false>);                 \
    /* The attribute holds ONE value per kernel function (the last set wins): keep the LARGEST size ever asked of a function and  \
     * set it again only when a larger one appears -- B = 32 (160 KB), then B = 64 (32 KB), then B = 32 again must still find 160 KB */ \
    static std::vector<std::pair<const void*, size_t>> attr_max;                                                 \
    {                                                                                                            \
      const void* kf = reinterpret_cast<const void*>(kfn);                                                       \
      auto it = std::find_if(attr_max.begin(), attr_max.end(), [&](const std::pair<const void*, size_t>& e) { return e.first == kf; }); \
      if (it == attr_max.end() || it->second < lds_launch) {                                                     \
        if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch) != hipSuccess) { (void)hipGetLastError(); } \
        if (it == attr_max.end()) attr_max.push_back({kf, lds_launch}); else it->second = lds_launch;           \
      }                                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, lds_launch, s, L);                                                      \
  } while (0)
    LAUNCH_C();
#undef LAUNCH_C
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mmda_set_error(bwd ? "mmda_lstm_bwd(cluster)" : "mmda_lstm_fwd(cluster)", e); return MMDA_ELAUNCH; }
  }
  *used = 1;
  return MMDA_OK;
}
