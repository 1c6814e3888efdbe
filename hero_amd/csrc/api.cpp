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
int ensure_dyn_lds(const void* fn, int bytes, std::atomic<uint64_t>& done, const char* what) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return HERO_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: device %d refused %d bytes of dynamic LDS: %s", what, dev, bytes, hipGetErrorString(e));
    return HERO_ERR_UNSUPPORTED;
  }
  done.fetch_or(bit, std::memory_order_release);
  return HERO_OK;
}
}  // namespace hero

extern "C" const char* hero_last_error(void) { return hero::g_err; }
extern "C" int hero_abi_version(void) { return HERO_ABI_VERSION; }
