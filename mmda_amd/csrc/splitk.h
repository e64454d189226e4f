// Deterministic split-K for the GEMMs (gemm.hip, gemm_bf16.hip): every K-slice of an output tile writes its raw fp32 partial tile
// into a slab of its own (per-stream scratch, api.hip), and ONE reduce launch behind the GEMM launch sums the slabs of each output
// element in slice order and applies the epilogue (alpha, bias, accumulate, gate interleave of the rows, bias-gradient column).
// Float atomics gave the same sums in arrival order: two identical steps differed in the last bits (reference train.py:46-51 asks
// for reproducible runs).
#pragma once
#include "common.h"

float* mmda_scratch_get(hipStream_t s, size_t bytes);       // api.hip

struct SplitKJob {
  const float* slab;        // [batch][sk][M][ldn] raw partial products
  float* C;                 // (M, ldc) per batch entry
  int M, N, ldn, ldc, sk, batch;
  int64_t strideC, strideBias;        // per batch entry (floats); strideBias also strides bias / bias_grad
  float alpha;
  const float* bias; const float* bias2;          // added once per element (index n; perm_n_H: at orig(n))
  float* bias_grad; float* bias_grad2;            // += column N of the slabs (index m; perm_m_H: at orig(m))
  int accumulate, perm_m_H, perm_n_H;
};
constexpr int SPLITK_JOBS_MAX = 16;

// the gate interleave of mmda_gemm_bf16_args: index j of the interleaved axis stands for torch's index orig(j)
__host__ __device__ __forceinline__ int splitk_orig(int j, int H) {
  const int G = 4 * H, d = j / G, r = j - d * G;
  return d * G + (r & 3) * H + (r >> 2);
}

int mmda_splitk_reduce(const SplitKJob* jobs, int n, hipStream_t s);   // splitk.hip
