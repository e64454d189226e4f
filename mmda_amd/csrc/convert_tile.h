// fp32 -> bf16 conversion / transposition tiles (device code shared by convert_kernel in gemm_bf16.hip and by the merged W_hh-packing +
// conversion launch in lstm.hip) and the host-side launch table.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int CONV_MAX = 16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// gate interleave: index j of an interleaved axis stands for torch's index orig(j) (see mmda_gemm_bf16_args)
__host__ __device__ __forceinline__ int gate_orig(int j, int H) {
  const int G = 4 * H, d = j / G, q = j - d * G;
  return d * G + (q & 3) * H + (q >> 2);
}

// ------------------------------------------------------------------------------------------------ conversion
// 64 x 64 tiles through LDS: coalesced fp32 reads along the source rows, coalesced bf16 writes along the rows of the plain
// copy and (transposed through LDS) along the rows of the transposed copy.  Padding columns up to the leading dimension are
// written as zeros so that the GEMM can read whole 16-byte chunks.
struct ConvLaunch {
  mmda_convert_job j[CONV_MAX];
  int start[CONV_MAX + 1];
  int tx[CONV_MAX];
  int vec[CONV_MAX];                                    // 16-byte form applicable (alignment / column count): convert_tile_vec
  int n;
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// 16-byte form of a 64 x 64 tile (job flag `vec`: 16-byte-aligned rows, column count a multiple of 4 [fp32 source] / 8 [bf16
// source]): every load is 16 bytes, the plain copy goes out as 8- / 16-byte stores straight from registers, and the transposed
// copy as 16-byte stores -- the tile is kept in LDS already transposed and packed, TT[column][row pair] = {row 2p, row 2p + 1},
// so eight consecutive rows of a column are four consecutive dwords.  The scalar form below moves 2 bytes per lane and store
// instruction; at B = 256 per GPU the five conversion launches of a step were 0.41 ms of it.
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void convert_tile_vec(const mmda_convert_job& J, int r0, int c0, unsigned* TT) {
  constexpr int TS = 33;                                 // dwords per TT row (conflict-light for both phases)
  const int tid = threadIdx.x;
  unsigned short* P = reinterpret_cast<unsigned short*>(J.plain);
  unsigned short* Tt = reinterpret_cast<unsigned short*>(J.transposed);
  auto src_row = [&](int r) -> int64_t {
    int rc = min(r, J.rows - 1);
    if (J.row_perm_H > 0) rc = gate_orig(rc, J.row_perm_H);
    return J.gather ? J.gather[rc] : (int64_t)rc;
  };
  if (J.src_bf16) {
    // item = (row pair rp, 8-column group c8): one per thread
    const int rp = tid >> 3, c8 = tid & 7;
    const int c = c0 + c8 * 8, cl = min(c, J.cols - 8);
    const unsigned short* s16 = reinterpret_cast<const unsigned short*>(J.src);
    u32x4 v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) v[e] = *reinterpret_cast<const u32x4*>(s16 + src_row(r0 + 2 * rp + e) * J.ld + cl);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = r0 + 2 * rp + e;
      if (!(r < J.rows && c < J.cols)) v[e] = u32x4{0u, 0u, 0u, 0u};
      if (P && r < J.rows && c < J.ldp) *reinterpret_cast<u32x4*>(P + (int64_t)r * J.ldp + c) = v[e];
    }
    if (Tt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {                      // columns 8 c8 + 2q, + 2q + 1
        TT[(c8 * 8 + 2 * q) * TS + rp] = (v[0][q] & 0xffffu) | (v[1][q] << 16);
        TT[(c8 * 8 + 2 * q + 1) * TS + rp] = (v[0][q] >> 16) | (v[1][q] & 0xffff0000u);
      }
    }
  } else {
    // item = (row pair rp, 4-column group c4): two per thread
    f32x4 v[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int id = tid + 256 * it, rp = id >> 4, c4 = id & 15;
      const int cl = min(c0 + c4 * 4, J.cols - 4);
#pragma unroll
      for (int e = 0; e < 2; ++e) v[it][e] = *reinterpret_cast<const f32x4*>(J.src + src_row(r0 + 2 * rp + e) * J.ld + cl);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int id = tid + 256 * it, rp = id >> 4, c4 = id & 15;
      const int c = c0 + c4 * 4;
      unsigned h[2][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = r0 + 2 * rp + e;
        const bool ok = r < J.rows && c < J.cols;
        h[e][0] = ok ? cvt_pk_bf16(v[it][e][0], v[it][e][1]) : 0u;
        h[e][1] = ok ? cvt_pk_bf16(v[it][e][2], v[it][e][3]) : 0u;
        if (P && r < J.rows && c < J.ldp) *reinterpret_cast<u32x2v*>(P + (int64_t)r * J.ldp + c) = u32x2v{h[e][0], h[e][1]};
      }
      if (Tt) {
        TT[(c4 * 4 + 0) * TS + rp] = (h[0][0] & 0xffffu) | (h[1][0] << 16);
        TT[(c4 * 4 + 1) * TS + rp] = (h[0][0] >> 16) | (h[1][0] & 0xffff0000u);
        TT[(c4 * 4 + 2) * TS + rp] = (h[0][1] & 0xffffu) | (h[1][1] << 16);
        TT[(c4 * 4 + 3) * TS + rp] = (h[0][1] >> 16) | (h[1][1] & 0xffff0000u);
      }
    }
  }
  if (!Tt) return;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {                       // 64 columns x 8 chunks of 8 rows
    const int id = tid + 256 * it, cc = id >> 3, j = id & 7;
    const int c = c0 + cc, r = r0 + 8 * j;
    if (c < J.cols && r < J.ldt) {
      const unsigned* t4 = TT + cc * TS + 4 * j;
      *reinterpret_cast<u32x4*>(Tt + (int64_t)c * J.ldt + r) = u32x4{t4[0], t4[1], t4[2], t4[3]};
    }
  }
}

// one 64 x 64 tile: `block` is the tile's index in the launch table (a kernel calls this with its own LDS tile)
__device__ __forceinline__ void convert_block(const ConvLaunch& L, int block, unsigned short (*tile)[66]) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < CONV_MAX; ++k)
    if (k < L.n && block >= L.start[k]) pi = k;
  const mmda_convert_job J = L.j[pi];
  const int local = block - L.start[pi];
  const int bx = local % L.tx[pi], by = local / L.tx[pi];
  const int r0 = by * 64, c0 = bx * 64;
  if (L.vec[pi]) {                                       // block-uniform
    convert_tile_vec(J, r0, c0, reinterpret_cast<unsigned*>(&tile[0][0]));
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  unsigned short* P = reinterpret_cast<unsigned short*>(J.plain);
  unsigned short* Tt = reinterpret_cast<unsigned short*>(J.transposed);
  // all 16 loads of a thread are issued from clamped addresses before the first use (no load under a lane-dependent branch)
  const int cc0 = min(c0 + tx, J.cols - 1);
  int64_t srow[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int rc = min(r0 + ty + 4 * i, J.rows - 1);
    if (J.row_perm_H > 0) rc = gate_orig(rc, J.row_perm_H);
    srow[i] = J.gather ? J.gather[rc] : (int64_t)rc;
  }
  unsigned short hv[16];
  if (J.src_bf16) {                                     // block-uniform: the source already holds bf16 (re-layout only)
    const unsigned short* s16 = reinterpret_cast<const unsigned short*>(J.src);
#pragma unroll
    for (int i = 0; i < 16; ++i) hv[i] = s16[srow[i] * J.ld + cc0];
  } else {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = J.src[srow[i] * J.ld + cc0];
#pragma unroll
    for (int i = 0; i < 16; ++i) hv[i] = f2bf(v[i]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rr = ty + 4 * i;
    const int r = r0 + rr, c = c0 + tx;
    const unsigned short h = (r < J.rows && c < J.cols) ? hv[i] : (unsigned short)0;
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


// fills the launch table from the jobs (validation included); returns MMDA_OK / MMDA_EINVAL and the number of tiles
inline int conv_build(const mmda_convert_job* jobs, int cnt, ConvLaunch& L, int& blocks) {
  L.n = 0;
  blocks = 0;
  for (int i = 0; i < cnt; ++i) {
    const mmda_convert_job& j = jobs[i];
    if (!j.src || j.rows < 0 || j.cols < 0 || (!j.plain && !j.transposed)) return MMDA_EINVAL;
    if (j.plain && (j.ldp < ((j.cols + 7) & ~7) || (j.ldp & 7))) return MMDA_EINVAL;
    if (j.transposed && (j.ldt < ((j.rows + 7) & ~7) || (j.ldt & 7))) return MMDA_EINVAL;
    if (j.row_perm_H < 0 || (j.row_perm_H && (j.gather || j.rows % (4 * j.row_perm_H)))) return MMDA_EINVAL;
    if (j.rows == 0 || j.cols == 0) continue;
    const int k = L.n++;
    L.j[k] = j;
    // tiles cover the padded extents so that the zero padding gets written
    const int ext_c = j.plain ? (j.ldp > j.cols ? j.ldp : j.cols) : j.cols;
    const int ext_r = j.transposed ? (j.ldt > j.rows ? j.ldt : j.rows) : j.rows;
    L.tx[k] = ceil_div(ext_c, 64);
    {
      static const int no_vec = getenv("MMDA_CONVERT_SCALAR") ? 1 : 0;      // ablation: the 2-byte-per-lane form
      const int g = j.src_bf16 ? 8 : 4;                // elements per 16-byte load
      bool v = !no_vec && (j.ld % g) == 0 && (j.cols % g) == 0 && j.cols >= g && ((uintptr_t)j.src & 15) == 0;
      if (j.plain) v = v && ((uintptr_t)j.plain & 15) == 0;
      if (j.transposed) v = v && ((uintptr_t)j.transposed & 15) == 0;
      L.vec[k] = v ? 1 : 0;
    }
    L.start[k] = blocks;
    blocks += L.tx[k] * ceil_div(ext_r, 64);
  }
  for (int k = L.n; k <= CONV_MAX; ++k) L.start[k] = blocks;
  for (int k = L.n; k < CONV_MAX; ++k) { L.j[k] = L.j[0]; L.tx[k] = 1; L.vec[k] = 0; }
  return MMDA_OK;
}

}  // namespace
