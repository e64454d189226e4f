// Error reporting + ABI version for libmmda_hip.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[256] = "";

void mmda_set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

extern "C" const char* mmda_last_error(void) { return g_err; }
extern "C" int mmda_abi_version(void) { return 1; }

// ---- per-stream scratch for the deterministic split-K of the GEMMs (slabs of partial tiles, summed in slice order by a reduce
// launch on the same stream).  One growable device buffer per stream handle: launches of one stream run in order, so the slabs of a
// grouped launch are consumed by its reduce launch before the next GEMM on that stream overwrites them; two streams never share a
// buffer.  Grown (hipMalloc: a device-wide synchronisation) only while the largest request is still being discovered, i.e. during
// warm-up; freed by mmda_scratch_release() or at process exit.
#include <mutex>
#include <unordered_map>
namespace {
struct Scratch { void* p = nullptr; size_t bytes = 0; };
std::mutex g_scratch_mu;
std::unordered_map<void*, Scratch> g_scratch;
}  // namespace

float* mmda_scratch_get(hipStream_t s, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  void* key = (void*)((uintptr_t)s ^ ((uintptr_t)(dev + 1) << 56));
  Scratch& sc = g_scratch[key];
  if (sc.bytes >= bytes && sc.p) return reinterpret_cast<float*>(sc.p);
  if (sc.p) {
    // work already queued on this stream may still read the old buffer
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    (void)hipFree(sc.p);
    sc.p = nullptr; sc.bytes = 0;
  }
  size_t want = bytes + bytes / 4;                 // head room: shapes that grow a little do not reallocate
  if (want < (size_t)8 << 20) want = (size_t)8 << 20;
  hipError_t e = hipMalloc(&sc.p, want);
  if (e != hipSuccess) { mmda_set_error("mmda_scratch_get", e); sc.p = nullptr; return nullptr; }
  sc.bytes = want;
  return reinterpret_cast<float*>(sc.p);
}

extern "C" int mmda_scratch_release(void) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  int rc = MMDA_OK;
  for (auto& kv : g_scratch)
    if (kv.second.p && hipFree(kv.second.p) != hipSuccess) rc = MMDA_ELAUNCH;
  g_scratch.clear();
  return rc;
}
