// Block-scaled fp8 (OCP MX: e4m3 elements, one E8M0 scale per 32 consecutive k) GEMM for the two feed-forward products of the
// fusion transformer layer -- linear1 (128 -> 2048) and linear2 (2048 -> 128), reference models.py:160-161 (BASELINE.json configs[4]:
// "mixed fp8 fusion GEMMs on CDNA4").  Off by default; mmda_misa_set_fusion_fp8 switches the FORWARD products over, the backward pass
// stays on the exact f32 path with the stored activations (straight-through).
//
// v_mfma_scale_f32_16x16x128_f8f6f4, operand layout as measured on gfx950 (tools/micro/mx_probe.hip, exact integer data):
//   lane l = (r = l & 15, g = l >> 4) holds 32 e4m3 bytes of row (A) / column (B) r:
//       bytes  0..15  <->  k = 16 g + j              (k in  0..63)
//       bytes 16..31  <->  k = 64 + 16 g + (j - 16)  (k in 64..127)
//   the E8M0 scale in byte 0 of lane (r, g)'s scale register applies to the MX block k in [32 g, 32 g + 32) of that row / column
//   C/D: col = lane & 15, row = 4 (lane >> 4) + reg  (as every 16x16 MFMA)
// The quantiser writes operands in exactly that order, so a GEMM lane fetches its 32 bytes with two 16-byte loads:
//   Q[row][kstep][g][32 bytes],  S[row][kstep] = one dword of four scale bytes (byte g = block g)
#include "common.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int QJOBS = 8;
struct QuantLaunch { mmda_mx8_quant_job j[QJOBS]; int start[QJOBS + 1]; int n; };

// one thread per (row, 32-element block): amax -> shared exponent (floor(log2 amax) - 8, the OCP MX rule for e4m3, emax = 8) ->
// elements x / 2^e clamped to +-448 and rounded to nearest even by v_cvt_pk_fp8_f32
__global__ __launch_bounds__(256) void mx8_quant_kernel(QuantLaunch L) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < QJOBS; ++k)
    if (k < L.n && (int)blockIdx.x >= L.start[k]) pi = k;
  const mmda_mx8_quant_job J = L.j[pi];
  const int nb = J.K / 32;                               // blocks per row
  const int64_t item = (int64_t)(blockIdx.x - L.start[pi]) * 256 + threadIdx.x;
  if (item >= (int64_t)J.rows * nb) return;
  const int row = (int)(item / nb), b = (int)(item % nb);
  const float* src = J.src + (int64_t)row * J.ld + b * 32;
  f4 v[8];
  if ((((uintptr_t)J.src) & 15) == 0 && (J.ld & 3) == 0) {          // job-uniform: 16-byte loads when the rows are aligned
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f4*>(src + 4 * i);
  } else {                                                          // (a parameter whose offset in the flat bucket is not a multiple of 4)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = src[4 * i + e];
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
  // floor(log2(amax)) from the exponent field (subnormal / zero amax: the smallest normal exponent; the elements then quantise to 0)
  const int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
  int se = (ex == 0 ? 1 : ex) - 8;                       // biased E8M0 scale: (floor(log2 amax) - 8) + 127
  se = se < 0 ? 0 : (se > 254 ? 254 : se);
  const float inv = __builtin_bit_cast(float, (unsigned)(254 - se) << 23);      // 2^-(se - 127), exact (se in 0..254 -> exponent 254..0)
  unsigned w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = fminf(fmaxf(v[i][e] * inv, -448.f), 448.f);
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], p, true);
    w[i] = (unsigned)p;
  }
  // block b = 4 ks + g of the row: its first 16 elements go to lane group gg0 = 2 (g & 1), its last 16 to gg0 + 1, both at bytes
  // 16 (g >> 1) .. + 15 of that group's 32 (see the operand layout above)
  const int ks = b >> 2, g = b & 3, gg0 = 2 * (g & 1), hb = g >> 1;
  unsigned char* q0 = J.q + (((int64_t)row * (nb / 4) + ks) * 4 + gg0) * 32 + hb * 16;
  *reinterpret_cast<u32x4*>(q0) = u32x4{w[0], w[1], w[2], w[3]};
  *reinterpret_cast<u32x4*>(q0 + 32) = u32x4{w[4], w[5], w[6], w[7]};
  J.s[(int64_t)row * nb + b] = (unsigned char)se;       // byte g of the dword S[row][ks]
}

struct Mx8Launch { mmda_mx8_args g; int nw; };

// One workgroup per 16 x 16 output tile; its nw waves take the k-steps (128 deep) round robin, partial tiles meet in LDS in a fixed
// order (bitwise reproducible), then the fused epilogue (bias, activation, dropout: same element index as the f32 path, m * N + n).
__global__ __launch_bounds__(1024) void gemm_mx8_kernel(Mx8Launch L) {
  __shared__ float red[16][256];
  const mmda_mx8_args& a = L.g;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tn = a.N / 16;
  const int bx = blockIdx.x % tn, by = blockIdx.x / tn;
  const int r = lane & 15, g = lane >> 4;
  const int KS = a.K / 128;
  const int arow = min(by * 16 + r, a.M - 1), bcol = bx * 16 + r;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ks = wave; ks < KS; ks += L.nw) {
    const unsigned char* pa = a.Aq + (((int64_t)arow * KS + ks) * 4 + g) * 32;
    const unsigned char* pb = a.Bq + (((int64_t)bcol * KS + ks) * 4 + g) * 32;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(pa), a1 = *reinterpret_cast<const u32x4*>(pa + 16);
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(pb), b1 = *reinterpret_cast<const u32x4*>(pb + 16);
    const unsigned sa = reinterpret_cast<const unsigned*>(a.As)[(int64_t)arow * KS + ks] >> (8 * g);
    const unsigned sb = reinterpret_cast<const unsigned*>(a.Bs)[(int64_t)bcol * KS + ks] >> (8 * g);
    const v8i av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const v8i bv = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, (int)(sa & 0xffu), 0, (int)(sb & 0xffu));
  }
  auto epilogue = [&](int m, int n, float v) {
    if (m >= a.M) return;
    if (a.bias) v += a.bias[n];
    v = act_fwd(a.act, v);
    if (a.drop_p > 0.f) v *= drop_mul(a.drop_p, a.drop_seed, a.drop_site, (uint64_t)m * a.N + n);
    a.C[(int64_t)m * a.ldc + n] = v;
  };
  if (L.nw == 1) {                                       // D fragment: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) epilogue(by * 16 + g * 4 + rg, bx * 16 + r, acc[rg]);
    return;
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) red[wave][(g * 4 + rg) * 16 + r] = acc[rg];
  __syncthreads();
  for (int e = threadIdx.x; e < 256; e += 64 * L.nw) {   // element (row e >> 4, column e & 15) of the tile
    float s = 0.f;
    for (int w = 0; w < L.nw; ++w) s += red[w][e];       // fixed order
    epilogue(by * 16 + (e >> 4), bx * 16 + (e & 15), s);
  }
}

}  // namespace

extern "C" int64_t mmda_mx8_quant_bytes(int rows, int K) {
  if (rows <= 0 || K <= 0 || (K % 128)) return MMDA_EINVAL;
  return (int64_t)rows * K;                              // element bytes; the scales take rows * K / 32 bytes more
}

extern "C" int mmda_mx8_quant(const mmda_mx8_quant_job* jobs, int n, void* stream) {
  if (!jobs || n < 0 || n > QJOBS) return MMDA_EINVAL;
  QuantLaunch L;
  L.n = 0;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const mmda_mx8_quant_job& j = jobs[i];
    if (!j.src || !j.q || !j.s || j.rows < 0 || j.K <= 0 || (j.K % 128) || j.ld < j.K) return MMDA_EINVAL;
    if (((uintptr_t)j.q & 15) || ((uintptr_t)j.s & 3)) return MMDA_EINVAL;
    if (j.rows == 0) continue;
    const int k = L.n++;
    L.j[k] = j;
    L.start[k] = blocks;
    blocks += (int)(((int64_t)j.rows * (j.K / 32) + 255) / 256);
  }
  for (int k = L.n; k <= QJOBS; ++k) L.start[k] = blocks;
  for (int k = L.n; k < QJOBS; ++k) L.j[k] = L.j[0];
  if (blocks == 0) return MMDA_OK;
  hipLaunchKernelGGL(mx8_quant_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L);
  MMDA_CHECK_LAUNCH("mmda_mx8_quant");
  return MMDA_OK;
}

extern "C" int mmda_gemm_mx8(const mmda_mx8_args* args, void* stream) {
  if (!args) return MMDA_EINVAL;
  const mmda_mx8_args& a = *args;
  if (!a.Aq || !a.As || !a.Bq || !a.Bs || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return MMDA_EINVAL;
  if ((a.K % 128) || (a.N % 16) || a.ldc < a.N) return MMDA_EINVAL;
  if ((((uintptr_t)a.Aq | (uintptr_t)a.Bq) & 15) || (((uintptr_t)a.As | (uintptr_t)a.Bs) & 3)) return MMDA_EINVAL;
  Mx8Launch L;
  L.g = a;
  const int KS = a.K / 128;
  L.nw = KS >= 16 ? 16 : (KS >= 8 ? 8 : (KS >= 4 ? 4 : (KS >= 2 ? 2 : 1)));
  const int tiles = (a.N / 16) * ceil_div(a.M, 16);
  hipLaunchKernelGGL(gemm_mx8_kernel, dim3(tiles), dim3(64 * L.nw), 0, (hipStream_t)stream, L);
  MMDA_CHECK_LAUNCH("mmda_gemm_mx8");
  return MMDA_OK;
}
