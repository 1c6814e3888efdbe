// Error plumbing shared by every translation unit of libhero_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

namespace hero {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return HERO_ERR_LAUNCH;
  }
  return HERO_OK;
}
}  // namespace hero

extern "C" const char* hero_last_error(void) { return hero::g_err; }
extern "C" int hero_abi_version(void) { return 1; }
