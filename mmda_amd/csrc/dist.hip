// Data-parallel pieces of the C ABI (the reference is single-device, its only trace of multi-GPU is the commented
// nn.DataParallel at solver.py:88-91):
//   * mmda_embed_segment_sum   deterministic sum of (id, row) pairs into the dense embedding gradient.  Ranks exchange the embedding
//                              gradient as all-gathered (ids, rows) -- at most T*B rows per rank instead of V rows -- and every rank
//                              then runs THIS on the same gathered list, in the same order, so the replicas stay bit-identical (the
//                              atomic scatter-add of mmda_embed_scatter_add sums duplicates in whatever order the hardware retires them).
//   * mmda_allreduce           in-place sum all-reduce of a float buffer on an RCCL communicator, for hosts that own one (a native
//                              trainer; the Python host goes through torch.distributed, whose communicator is not exposed).
#include "common.h"
#include "splitk.h"
#include <dlfcn.h>
#include <hipcub/hipcub.hpp>

namespace {

constexpr int CHUNK = 64;          // run boundaries = segment boundaries U multiples of CHUNK in the sorted list

__global__ void seg_keys_kernel(const int64_t* __restrict__ ids, int n, unsigned* keys, int* vals, const int* __restrict__ lengths, int B,
                                unsigned pad_key) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int64_t id = ids[p];
  // negative ids (padding markers of the gathered lists) and, with `lengths`, the positions p = t * B + b past a sample's length
  // (their rows are exactly zero) sort to the end and are skipped
  const bool pad = id < 0 || (lengths != nullptr && (p / B) >= lengths[p % B]);
  keys[p] = (pad || (unsigned)id >= pad_key) ? pad_key : (unsigned)id;      // (pad_key: above every id; ids beyond it cannot be table rows)
  vals[p] = p;
}

__device__ __forceinline__ bool run_start(const unsigned* sid, int p) { return (p % CHUNK) == 0 || sid[p] != sid[p - 1]; }

// level 1: one wave per run start; part[p] = rows[pos[p]] + rows[pos[p+1]] + ... over the run, in list order.  A run that is its
// id's WHOLE segment (nearly all of them: a segment spans runs only across a chunk boundary) goes straight into the table row --
// the same bits level 2 would have produced from the one partial (0 + part, then the row's update) without the trip through `part`.
__global__ __launch_bounds__(256) void seg_level1_kernel(const unsigned* __restrict__ sid, const int* __restrict__ pos, int n, int D,
                                                         const float* __restrict__ rows, float* part, unsigned pad_key, float* dW,
                                                         int accumulate) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= n) return;
  const unsigned id = sid[p];
  if (id == pad_key || !run_start(sid, p)) return;                 // wave-uniform
  int end = p + 1;
  const int lim = min(n, (p / CHUNK + 1) * CHUNK);
  while (end < lim && sid[end] == id) ++end;
  const bool whole = (p == 0 || sid[p - 1] != id) && (end == n || sid[end] != id);      // wave-uniform
  for (int c = lane; c < D; c += 64) {
    float acc = 0.f;
    int q = p;
    for (; q + 4 <= end; q += 4) {                                   // four independent loads in flight, added in list order
      const float a = rows[(int64_t)pos[q] * D + c], b = rows[(int64_t)pos[q + 1] * D + c];
      const float e = rows[(int64_t)pos[q + 2] * D + c], f = rows[(int64_t)pos[q + 3] * D + c];
      acc += a; acc += b; acc += e; acc += f;
    }
    for (; q < end; ++q) acc += rows[(int64_t)pos[q] * D + c];
    if (whole) {
      float* dst = dW + (int64_t)id * D + c;                          // one writer per table row
      *dst = accumulate ? *dst + acc : acc;
    } else {
      part[(int64_t)p * D + c] = acc;
    }
  }
}

// level 2: one wave per segment head; dW[id] = sum of the segment's run partials, in list order (overwrites the row)
__global__ __launch_bounds__(256) void seg_level2_kernel(const unsigned* __restrict__ sid, int n, int D, const float* __restrict__ part,
                                                         float* dW, int accumulate, unsigned pad_key) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= n) return;
  const unsigned id = sid[p];
  if (id == pad_key || (p > 0 && sid[p - 1] == id)) return;         // wave-uniform: not a segment head
  {
    const int q2 = (p / CHUNK + 1) * CHUNK;                          // a single run: level 1 wrote the row itself
    if (q2 >= n || sid[q2] != id) return;
  }
  for (int c = lane; c < D; c += 64) {
    float acc = 0.f;
    for (int q = p; q < n && sid[q] == id; q = (q / CHUNK + 1) * CHUNK) acc += part[(int64_t)q * D + c];
    float* dst = dW + (int64_t)id * D + c;                            // one writer per table row
    *dst = accumulate ? *dst + acc : acc;
  }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct SegLayout { size_t keys_in, keys_out, vals_in, vals_out, part, cub, total, cub_bytes; };
SegLayout seg_layout(int n, int D) {
  SegLayout L{};
  size_t o = 0;
  L.keys_in = o; o += align256(sizeof(unsigned) * (size_t)n);
  L.keys_out = o; o += align256(sizeof(unsigned) * (size_t)n);
  L.vals_in = o; o += align256(sizeof(int) * (size_t)n);
  L.vals_out = o; o += align256(sizeof(int) * (size_t)n);
  L.part = o; o += align256(sizeof(float) * (size_t)n * D);
  size_t cb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs((void*)nullptr, cb, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                           (int*)nullptr, n, 0, 32, (hipStream_t)0);
  L.cub_bytes = cb;
  L.cub = o; o += align256(cb);
  L.total = o;
  return L;
}

}  // namespace

extern "C" int64_t mmda_embed_segment_sum_work_bytes(int n, int D) {
  if (n < 0 || D <= 0) return MMDA_EINVAL;
  if (n == 0) return 256;
  return (int64_t)seg_layout(n, D).total;
}

static int seg_sum(float* dW, const int64_t* ids, int n, int D, const float* rows, void* work, int64_t work_bytes, void* stream, int accumulate,
                   const int* lengths = nullptr, int B = 0);

extern "C" int mmda_embed_segment_sum(float* dW, const int64_t* ids, int n, int D, const float* rows, void* work, int64_t work_bytes,
                                      void* stream) {
  return seg_sum(dW, ids, n, D, rows, work, work_bytes, stream, 0);
}

// dW[ids[p]] += rows[p] by the same machinery (stable sort, list-order sums), work buffer from the stream's scratch: what
// mmda_embed_scatter_add runs for long lists, where its one-workgroup-per-position scan (O(n^2 / 256) id compares) loses
int mmda_embed_scatter_sorted(float* dW, const int64_t* ids, int n, int D, const float* rows, const int* lengths, int B, void* stream) {
  if (n == 0) return MMDA_OK;
  const int64_t bytes = (int64_t)seg_layout(n, D).total;
  float* work = mmda_scratch_get((hipStream_t)stream, (size_t)bytes + 256);
  if (!work) return MMDA_ELAUNCH;
  void* w = (void*)(((uintptr_t)work + 255) & ~(uintptr_t)255);
  return seg_sum(dW, ids, n, D, rows, w, bytes, stream, 1, lengths, B);
}

// the sorted (key, position) list of an id list: keys_out / vals_out (n words each); kin / vin / cub: temporaries.  `bits`: key width
// the sort walks (32, or what holds pad_key when the caller knows the table's row count: two 8-bit passes instead of four)
static int seg_sort(const int64_t* ids, int n, const int* lengths, int B, unsigned pad_key, int bits, unsigned* kin, int* vin, unsigned* kout,
                    int* vout, void* cub, size_t cub_bytes, hipStream_t s) {
  hipLaunchKernelGGL(seg_keys_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, ids, n, kin, vin, lengths, B, pad_key);
  MMDA_CHECK_LAUNCH("mmda_embed_segment_sum/keys");
  size_t cb = cub_bytes;
  // LSD radix sort: stable, so equal ids keep their list order (the order every rank sums them in)
  if (hipcub::DeviceRadixSort::SortPairs(cub, cb, kin, kout, vin, vout, n, 0, bits, s) != hipSuccess) {
    mmda_set_error("mmda_embed_segment_sum/sort", hipGetLastError());
    return MMDA_ELAUNCH;
  }
  return MMDA_OK;
}
static int seg_reduce(const unsigned* kout, const int* vout, int n, int D, const float* rows, float* part, float* dW, int accumulate,
                      unsigned pad_key, hipStream_t s) {
  hipLaunchKernelGGL(seg_level1_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, kout, vout, n, D, rows, part, pad_key, dW, accumulate);
  MMDA_CHECK_LAUNCH("mmda_embed_segment_sum/level1");
  hipLaunchKernelGGL(seg_level2_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, kout, n, D, part, dW, accumulate, pad_key);
  MMDA_CHECK_LAUNCH("mmda_embed_segment_sum/level2");
  return MMDA_OK;
}

static int seg_sum(float* dW, const int64_t* ids, int n, int D, const float* rows, void* work, int64_t work_bytes, void* stream, int accumulate,
                   const int* lengths, int B) {
  if (!dW || !ids || !rows || !work || n < 0 || D <= 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  const SegLayout L = seg_layout(n, D);
  if (work_bytes < (int64_t)L.total || ((uintptr_t)work & 255)) return MMDA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* w = (unsigned char*)work;
  unsigned* kin = (unsigned*)(w + L.keys_in); unsigned* kout = (unsigned*)(w + L.keys_out);
  int* vin = (int*)(w + L.vals_in); int* vout = (int*)(w + L.vals_out);
  float* part = (float*)(w + L.part);
  int rc = seg_sort(ids, n, lengths, B, 0xFFFFFFFFu, 32, kin, vin, kout, vout, (void*)(w + L.cub), L.cub_bytes, s);
  if (rc) return rc;
  return seg_reduce(kout, vout, n, D, rows, part, dW, accumulate, 0xFFFFFFFFu, s);
}

// ---- the two halves on their own (internal, misa.hip): the sorted list depends on the ids only, so a training step makes it early,
// beside a recurrence, and only the sums wait for the gradient rows.  `sorted`: 2 n words the caller keeps (keys, then positions);
// table_rows: ids lie in [0, table_rows) -- the sort then walks the bits of table_rows only.
int mmda_embed_sort_ids(const int64_t* ids, int n, const int* lengths, int B, int table_rows, unsigned* sorted, void* stream) {
  if (!ids || !sorted || n < 0 || table_rows <= 0 || (lengths && B <= 0)) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  const SegLayout L = seg_layout(n, 1);
  float* work = mmda_scratch_get((hipStream_t)stream, L.total + 256);
  if (!work) return MMDA_ELAUNCH;
  unsigned char* w = (unsigned char*)(((uintptr_t)work + 255) & ~(uintptr_t)255);
  int bits = 1;
  while (bits < 32 && ((unsigned)table_rows >> bits) != 0u) ++bits;      // table_rows (the pad key) < 2^bits
  return seg_sort(ids, n, lengths, B, (unsigned)table_rows, bits, (unsigned*)(w + L.keys_in), (int*)(w + L.vals_in), sorted,
                  reinterpret_cast<int*>(sorted + n), (void*)(w + L.cub), L.cub_bytes, (hipStream_t)stream);
}
int mmda_embed_scatter_presorted(float* dW, const unsigned* sorted, int n, int D, int table_rows, const float* rows, void* stream) {
  if (!dW || !sorted || !rows || n < 0 || D <= 0 || table_rows <= 0) return MMDA_EINVAL;
  if (n == 0) return MMDA_OK;
  float* part = mmda_scratch_get((hipStream_t)stream, sizeof(float) * (size_t)n * D);
  if (!part) return MMDA_ELAUNCH;
  return seg_reduce(sorted, reinterpret_cast<const int*>(sorted + n), n, D, rows, part, dW, 1, (unsigned)table_rows, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- RCCL all-reduce
// The entry points are looked up at run time: first among the symbols already loaded in the process (the library that created the
// caller's communicator -- a communicator is only valid inside the library instance it came from), then in librccl.so.  The shared
// library therefore carries no link-time dependency on RCCL and loads on machines without it.
namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
nccl_allreduce_fn find_allreduce() {
  static nccl_allreduce_fn fn = nullptr;
  static bool looked = false;
  if (looked) return fn;
  looked = true;
  void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
  if (!sym) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) sym = dlsym(h, "ncclAllReduce");
  }
  fn = (nccl_allreduce_fn)sym;
  return fn;
}
}  // namespace

extern "C" int mmda_allreduce(void* buf, size_t n_floats, void* nccl_comm, void* stream) {
  if (!buf || !nccl_comm) return MMDA_EINVAL;
  if (n_floats == 0) return MMDA_OK;
  nccl_allreduce_fn ar = find_allreduce();
  if (!ar) { mmda_set_error("mmda_allreduce: ncclAllReduce not found (librccl.so)", hipErrorNotFound); return MMDA_ELAUNCH; }
  // ncclFloat32 = 7, ncclSum = 0 (rccl.h); in place
  const int rc = ar(buf, buf, n_floats, 7, 0, nccl_comm, (hipStream_t)stream);
  if (rc != 0) { mmda_set_error("mmda_allreduce: ncclAllReduce failed", hipErrorUnknown); return MMDA_ELAUNCH; }
  return MMDA_OK;
}
