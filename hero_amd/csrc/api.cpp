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
extern "C" int hero_abi_struct_count(void) { return HERO_STRUCT_COUNT_; }
extern "C" int hero_abi_struct_bytes(int which) {
  switch (which) {
    case HERO_STRUCT_DROPOUT: return (int)sizeof(HeroDropout);
    case HERO_STRUCT_GEMM_EPILOGUE: return (int)sizeof(HeroGemmEpilogue);
    case HERO_STRUCT_WGRAD_PROBLEM: return (int)sizeof(HeroWgradProblem);
    case HERO_STRUCT_LN_FWD: return (int)sizeof(HeroLnFwd);
    case HERO_STRUCT_LN_BWD: return (int)sizeof(HeroLnBwd);
    case HERO_STRUCT_COLSUM: return (int)sizeof(HeroColsum);
    case HERO_STRUCT_ATTN: return (int)sizeof(HeroAttn);
    case HERO_STRUCT_ADAMW: return (int)sizeof(HeroAdamW);
    case HERO_STRUCT_TENSOR_DESC: return (int)sizeof(HeroTensorDesc);
    case HERO_STRUCT_ADAMW_GROUP: return (int)sizeof(HeroAdamWGroup);
    case HERO_STRUCT_ADAMW_MULTI: return (int)sizeof(HeroAdamWMulti);
    case HERO_STRUCT_COPY_DESC: return (int)sizeof(HeroCopyDesc);
    case HERO_STRUCT_QUERY_POOL: return (int)sizeof(HeroQueryPool);
    case HERO_STRUCT_ROW_NORM: return (int)sizeof(HeroRowNorm);
    case HERO_STRUCT_SCORE_MAX: return (int)sizeof(HeroScoreMax);
    case HERO_STRUCT_RANK_LOSS: return (int)sizeof(HeroRankLoss);
    case HERO_STRUCT_ST_ED: return (int)sizeof(HeroStEd);
    case HERO_STRUCT_CROSS_ENTROPY: return (int)sizeof(HeroCrossEntropy);
    case HERO_STRUCT_DERIVE: return (int)sizeof(HeroDerive);
    case HERO_STRUCT_COMM_BUCKET: return (int)sizeof(HeroCommBucket);
    default: return -1;
  }
}
