// Shared device helpers for the hero_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/hero_hip.h"

namespace hero {

typedef uint16_t bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// float -> bf16 is the hardware conversion (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN kept)
typedef __bf16 hwbf2_t __attribute__((ext_vector_type(2)));
typedef float hwf2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
  const hwf2_t v = {lo, hi};
  const hwbf2_t b = __builtin_convertvector(v, hwbf2_t);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf_pk(f, 0.f) & 0xffffu); }

// ---- 4-element vector access in the activation dtype ----------------------------------------
template <typename T> struct V4;
template <> struct V4<float> {
  static __device__ __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct V4<bf16_t> {
  static __device__ __forceinline__ float4 ld(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  }
  static __device__ __forceinline__ void st(bf16_t* p, float4 v) {
    uint2 u;
    u.x = f2bf_pk(v.x, v.y);
    u.y = f2bf_pk(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
  }
};
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// ---- counter-based dropout -------------------------------------------------------------------
// The keep decisions of 4 consecutive elements (a "group") are four 16-bit uniforms: two 32-bit hashes of the group
// index, keyed by the two halves of splitmix64(seed word ^ site).  The same (seed word, site, group index) is replayed
// by the backward kernels, so no mask is stored.
// The per-group hash is mix32 (C. Wellons' "lowbias32": xorshift-multiply x 2, avalanche bias 0.17) - 4 quarter-rate
// 32-bit multiplies per group.  Rounds 1-2 drew the 64 bits from splitmix64(base + group * C): three 64 x 64-bit
// multiplies = 12 quarter-rate multiplies + 64-bit shifts per group.  The 32-bit form has ~40 % fewer VALU slots per draw
// and the step did not notice (6.63 vs 6.61 ms): what the masks cost in the kernels (-DHERO_DROP_ABLATE, rocprofv3 A/B:
// 0.06 ms per micro-step - LayerNorm backward 1.2 us per launch, the dropout GEMM epilogue 2.2 us, attention 0.7-1 us)
// is the index / compare / select work around the draw, not its multiplies.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  return x ^ (x >> 16);
}
struct DropCtx {
  uint32_t k0, k1;
  uint32_t thr;
  float scale;
  __device__ __forceinline__ DropCtx(const HeroDropout& d) {
    thr = d.threshold16;
    scale = d.scale;
    const uint64_t base = thr ? splitmix64((d.seed_ptr ? *d.seed_ptr : 0ull) ^ (d.site * 0xD6E8FEB86659FD93ull)) : 0ull;
    k0 = (uint32_t)base;
    k1 = (uint32_t)(base >> 32);
  }
  __device__ __forceinline__ bool on() const { return thr != 0; }
  // group index -> 32-bit hash input (the high word, zero below 2^34 elements, is folded in with a full-rate 24-bit multiply)
  __device__ __forceinline__ uint32_t fold(uint64_t group) const {
    return (uint32_t)group ^ __umul24((uint32_t)(group >> 32), 0xC2B2AEu);
  }
  // group = index of a 4-element group; returns per-element multipliers (0 or scale)
  __device__ __forceinline__ float4 mask4(uint64_t group) const {
#ifdef HERO_DROP_ABLATE       // lab only: what the hash costs (keeps everything)
    return make_float4(scale, scale, scale, scale);
#endif
    const uint32_t g = fold(group);
    const uint32_t r0 = mix32(g ^ k0), r1 = mix32(g ^ k1);
    float4 m;
    m.x = ((r0 & 0xffffu) >= thr) ? scale : 0.f;
    m.y = ((r0 >> 16) >= thr) ? scale : 0.f;
    m.z = ((r1 & 0xffffu) >= thr) ? scale : 0.f;
    m.w = ((r1 >> 16) >= thr) ? scale : 0.f;
    return m;
  }
  __device__ __forceinline__ float mask1(uint64_t elem) const {
    const uint32_t r = mix32(fold(elem >> 2) ^ ((elem & 2) ? k1 : k0));
    return (((r >> (16 * (uint32_t)(elem & 1))) & 0xffffu) >= thr) ? scale : 0.f;
  }
};

// Wave-wide reductions on the DPP network (no LDS round trips): row_shr 1/2/4/8 leave each 16-lane
// row's total in its lane 15, row_bcast:15 / row_bcast:31 fold the four rows into lane 63, which
// is read back through an SGPR.  The result is uniform across the wave.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float identity, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0x111, 0xf, 0xf>(0.f, v);
  v += dpp_mov<0x112, 0xf, 0xf>(0.f, v);
  v += dpp_mov<0x114, 0xf, 0xe>(0.f, v);
  v += dpp_mov<0x118, 0xf, 0xc>(0.f, v);
  v += dpp_mov<0x142, 0xa, 0xf>(0.f, v);
  v += dpp_mov<0x143, 0xc, 0xf>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float lo = -3.0e38f;
  v = fmaxf(v, dpp_mov<0x111, 0xf, 0xf>(lo, v));
  v = fmaxf(v, dpp_mov<0x112, 0xf, 0xf>(lo, v));
  v = fmaxf(v, dpp_mov<0x114, 0xf, 0xe>(lo, v));
  v = fmaxf(v, dpp_mov<0x118, 0xf, 0xc>(lo, v));
  v = fmaxf(v, dpp_mov<0x142, 0xa, 0xf>(lo, v));
  v = fmaxf(v, dpp_mov<0x143, 0xc, 0xf>(lo, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// sum over the 2 or 4 adjacent lanes that share a key (quad_perm swaps)
template <int LPK> __device__ __forceinline__ float group_sum(float v) {
  if (LPK >= 2) v += dpp_mov<0xB1, 0xf, 0xf>(0.f, v);   // quad_perm [1,0,3,2]
  if (LPK >= 4) v += dpp_mov<0x4E, 0xf, 0xf>(0.f, v);   // quad_perm [2,3,0,1]
  return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// GELU in the bf16 compute mode: Phi(x) from the Abramowitz-Stegun 7.1.26 erfc form,
//   erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2), t = 1/(1 + p z), |abs err| <= 1.5e-7,
// with z = |x|/sqrt(2) so that exp(-z^2) = exp(-x^2/2) is ALSO the Gaussian of gelu'.  About 15 VALU
// instructions (one v_exp, one v_rcp) against ~60 for erff: the libm erf made the GELU epilogues of
// the two 3072-wide GEMMs VALU-bound (+56 us on an 82 us GEMM).  The error is 2^-23-class, three
// orders of magnitude below the bf16 rounding of the stored result; the f32 parity mode keeps erff.
struct GeluFast {
  float cdf, q;    // Phi(x), exp(-x^2/2)
  __device__ __forceinline__ GeluFast(float x) {
    const float ax = fabsf(x);
    q = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);           // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float half_erfc = 0.5f * p * t * q;                               // 0.5 erfc(|x|/sqrt2)
    cdf = x >= 0.f ? 1.f - half_erfc : half_erfc;
  }
};
template <typename T> __device__ __forceinline__ float gelu_fwd(float x);
template <> __device__ __forceinline__ float gelu_fwd<float>(float x) { return gelu_erf(x); }
#ifdef HERO_GELU_ABLATE        // lab only (tools/lab/build_variants.sh): what the GELU arithmetic costs in the epilogues
template <> __device__ __forceinline__ float gelu_fwd<bf16_t>(float x) { return x * 0.5f; }
#else
template <> __device__ __forceinline__ float gelu_fwd<bf16_t>(float x) { return x * GeluFast(x).cdf; }
#endif
template <typename T> __device__ __forceinline__ float gelu_grad(float x);
template <> __device__ __forceinline__ float gelu_grad<float>(float x) { return gelu_erf_grad(x); }
template <> __device__ __forceinline__ float gelu_grad<bf16_t>(float x) {
#ifdef HERO_GELU_ABLATE
  return 0.5f;
#endif
  const GeluFast g(x);
  return fmaf(x * 0.39894228040143267794f, g.q, g.cdf);
}

// gelu(x) and gelu'(x) from ONE evaluation of Phi / the Gaussian (HERO_ACT_GELU_DG)
template <typename T> __device__ __forceinline__ void gelu_both(float x, float& y, float& dy);
template <> __device__ __forceinline__ void gelu_both<float>(float x, float& y, float& dy) { y = gelu_erf(x); dy = gelu_erf_grad(x); }
template <> __device__ __forceinline__ void gelu_both<bf16_t>(float x, float& y, float& dy) {
  const GeluFast g(x);
  y = x * g.cdf;
  dy = fmaf(x * 0.39894228040143267794f, g.q, g.cdf);
}

// error plumbing (api.cpp)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// "This kernel may use `bytes` of dynamic LDS": hipFuncSetAttribute is a PER-DEVICE setting and can be refused, so it is
// applied once per (kernel, device) - a bit per device ordinal in the call site's own atomic mask, thread-safe - and its
// result is checked (the results were discarded behind process-wide `static bool` flags).
int ensure_dyn_lds(const void* fn, int bytes, std::atomic<uint64_t>& done, const char* what);
#define HERO_ENSURE_LDS(fnptr, bytes, what)                                                                     \
  do {                                                                                                          \
    static std::atomic<uint64_t> lds_ok_{0};                                                                    \
    const int lds_rc_ = ::hero::ensure_dyn_lds(reinterpret_cast<const void*>(fnptr), (int)(bytes), lds_ok_, (what)); \
    if (lds_rc_ != HERO_OK) return lds_rc_;                                                                     \
  } while (0)

#define HERO_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      ::hero::set_error(__VA_ARGS__);  \
      return HERO_ERR_ARG;             \
    }                                  \
  } while (0)

}  // namespace hero
