// Shared device/host helpers for the gfx950 kernels (wave64, MFMA fragment types, RNG, reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mmda_hip.h"

#define WAVE 64

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 accumulator fragment

void mmda_set_error(const char* what, hipError_t e);
#define MMDA_CHECK_LAUNCH(name)                                   \
  do {                                                            \
    hipError_t _e = hipGetLastError();                            \
    if (_e != hipSuccess) { mmda_set_error(name, _e); return MMDA_ELAUNCH; } \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// fp32 -> bf16 round-to-nearest-even via the hardware cast (keeps NaN a NaN, MI355X_MICROARCH correctness table)
__device__ __forceinline__ unsigned short f2bf(float x) {
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
  return __builtin_bit_cast(float, ((unsigned)u) << 16);
}

// accurate versions (fp32 parity path and every non-recurrent kernel)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh via exp(-2|x|): saturates cleanly, |err| ~1e-7
__device__ __forceinline__ float tanhf_(float x) {
  float ax = fabsf(x);
  float e = expf(-2.0f * ax);
  float t = (1.0f - e) / (1.0f + e);
  return copysignf(t, x);
}
// fast versions (v_exp_f32 + v_rcp_f32) for the bf16 recurrent kernels where the gate math sits on the serial chain
// (__frcp_rn is a correctly rounded reciprocal = a ~10-instruction division sequence; v_rcp_f32 is 1 ulp and one issue)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// Counter-based dropout RNG: murmur3 fmix64 of (seed, site, element index).  The same triple gives the same bit in
// forward and backward, so masks are never stored.
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, int site, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1) + idx * 0xD6E8FEB86659FD93ull;
  z ^= z >> 33; z *= 0xff51afd7ed558ccdull;
  z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull;
  z ^= z >> 33;
  return (uint32_t)(z >> 16);
}
// inverted-dropout multiplier: 0 with probability p, else 1/(1-p)
__device__ __forceinline__ float drop_mul(float p, uint64_t seed, int site, uint64_t idx) {
  if (p <= 0.0f) return 1.0f;
  uint32_t r = rng_u32(seed, site, idx);
  float u = (float)(r >> 8) * (1.0f / 16777216.0f);
  return u < p ? 0.0f : 1.0f / (1.0f - p);
}

// Wave-wide sum / max, the same value in every lane.  Inside a row of 16 lanes by DPP (quad swaps, half-row mirror, row mirror: 4
// VALU instructions, every lane of a row ends with the row's result); the four rows are then read as scalars and combined.  (A
// __shfl_xor butterfly is six dependent ds_bpermute round trips, ~700 cycles: the 36 dot products of the six-token attention cost 11 us
// that way.)
#define MMDA_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum(float v) {
  v += MMDA_DPP(v, 0xB1);            // quad_perm [1,0,3,2]
  v += MMDA_DPP(v, 0x4E);            // quad_perm [2,3,0,1]
  v += MMDA_DPP(v, 0x141);           // row_half_mirror
  v += MMDA_DPP(v, 0x140);           // row_mirror
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, MMDA_DPP(v, 0xB1));
  v = fmaxf(v, MMDA_DPP(v, 0x4E));
  v = fmaxf(v, MMDA_DPP(v, 0x141));
  v = fmaxf(v, MMDA_DPP(v, 0x140));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case MMDA_ACT_RELU: return x > 0.f ? x : 0.f;
    case MMDA_ACT_SIGMOID: return sigmoidf_(x);
    case MMDA_ACT_LEAKYRELU: return x > 0.f ? x : 0.01f * x;
    case MMDA_ACT_TANH: return tanhf_(x);
    case MMDA_ACT_ELU: return x > 0.f ? x : (__expf(x) - 1.0f);
    case MMDA_ACT_HARDTANH: return fminf(fmaxf(x, -1.f), 1.f);
    case MMDA_ACT_HARDSHRINK: return (x > 0.5f || x < -0.5f) ? x : 0.f;
    default: return x;
  }
}
// derivative w.r.t. the pre-activation x
__device__ __forceinline__ float act_bwd(int act, float x) {
  switch (act) {
    case MMDA_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case MMDA_ACT_SIGMOID: { float s = sigmoidf_(x); return s * (1.f - s); }
    case MMDA_ACT_LEAKYRELU: return x > 0.f ? 1.f : 0.01f;
    case MMDA_ACT_TANH: { float t = tanhf_(x); return 1.f - t * t; }
    case MMDA_ACT_ELU: return x > 0.f ? 1.f : __expf(x);
    case MMDA_ACT_HARDTANH: return (x > -1.f && x < 1.f) ? 1.f : 0.f;
    case MMDA_ACT_HARDSHRINK: return (x > 0.5f || x < -0.5f) ? 1.f : 0.f;
    default: return 1.f;
  }
}

// ---- parametrised activations (nn.PReLU / nn.RReLU of the reference's activation_dict, config.py:25-27)
__device__ __forceinline__ float act_slope_p(int act, const mmda_act_params& p, uint64_t idx) {
  if (act == MMDA_ACT_PRELU) return p.slope[0];
  if (!p.rand) return 0.5f * (p.lo + p.hi);
  const float u = (float)(rng_u32(p.seed, p.site, idx) >> 8) * (1.0f / 16777216.0f);
  return p.lo + (p.hi - p.lo) * u;
}
__device__ __forceinline__ bool act_is_p(int act) { return act == MMDA_ACT_PRELU || act == MMDA_ACT_RRELU; }
__device__ __forceinline__ float act_fwd_p(int act, float x, const mmda_act_params& p, uint64_t idx) {
  if (!act_is_p(act)) return act_fwd(act, x);
  return x > 0.f ? x : act_slope_p(act, p, idx) * x;
}
__device__ __forceinline__ float act_bwd_p(int act, float x, const mmda_act_params& p, uint64_t idx) {
  if (!act_is_p(act)) return act_bwd(act, x);
  return x > 0.f ? 1.f : act_slope_p(act, p, idx);
}

// internal: two buffers cleared by one launch (optim.hip); the tail flag of the single-workgroup CMD launch (losses.hip)
int mmda_zero2(float* a, int64_t na, float* b, int64_t nb, void* stream);
bool mmda_loss_cmd_sets_flag(int B, int D);
void mmda_loss_cmd_arm_flag(unsigned* flag, unsigned value);
// internal: mmda_clamp_adam whose launch does not complete before *wait_flag reaches wait_value (optim.hip)
int mmda_clamp_adam_wait(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float clip,
                         float grad_scale, int step, const unsigned* wait_flag, unsigned wait_value, unsigned* wait_err, void* stream);

// ---- flag joins (misa.hip: side_flag_signal): a kernel of one stream waits, on the device, for a word that a one-thread launch behind
// the last kernel of ANOTHER stream's chain sets to `value` -- instead of a stream-level event wait, which costs the waiting stream
// 9 - 12 us of packet processing however early the other chain finished (tools/micro/fork_cost.hip).  Called by every thread of the
// workgroup; thread 0 polls (bounded: a word that never arrives is reported through *err and the kernel goes on), a workgroup
// barrier and an agent-scope acquire follow.  flag == nullptr: nothing to wait for (workgroup-uniform).
__device__ __forceinline__ void flag_wait(const unsigned* flag, unsigned value, unsigned* err) {
  if (!flag) return;
  if (threadIdx.x == 0) {
    int polls = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
      __builtin_amdgcn_s_sleep(16);
      if (++polls > (1 << 21)) { if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

