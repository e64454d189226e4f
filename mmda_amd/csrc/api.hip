// Error reporting + ABI version for libmmda_hip.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[256] = "";

void mmda_set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

extern "C" const char* mmda_last_error(void) { return g_err; }
extern "C" int mmda_abi_version(void) { return 1; }
