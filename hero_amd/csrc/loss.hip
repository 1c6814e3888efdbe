// Row-wise softmax cross-entropy over wide logit rows, forward and backward (gfx950).
//
// Replaces F.cross_entropy at the pre-training heads (BASELINE configs[3]):
//   MLM     model/encoder.py:355-389 + model/layers.py:330-354: [n_masked, 50272] logits of the tied-weight
//           vocabulary GEMM (the last `pad` columns are vocabulary padding, model/encoder.py:232-233: excluded)
//   MFM-NCE model/model.py:271-291: [n_masked, n_masked + n_neg] logits / temperature
//   FOM     model/model.py:293-336: [B * L, max_clip_len] logits, ignore_index -1
// HBM-bound: forward reads the logits once (online max / sum per thread, one block per row), backward reads
// them once more and writes the gradient in the logits' dtype (may alias the logits):
//   loss_r = lse_r - x[r, y_r],   dx[r, c] = (exp(x[r, c] - lse_r) - [c == y_r]) * g_r * inv_temp
// with x = logits * inv_temp; rows whose label equals ignore_index give loss 0 and a zero gradient row.
#include "common.h"

namespace hero {
namespace {

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = f2bf_pk(v[0], v[1]); u.y = f2bf_pk(v[2], v[3]); u.z = f2bf_pk(v[4], v[5]); u.w = f2bf_pk(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// (max, sum exp(x - max)) pairs combine associatively
__device__ __forceinline__ void ms_merge(float& m, float& s, float m2, float s2) {
  const float mx = fmaxf(m, m2);
  s = s * __expf(m - mx) + s2 * __expf(m2 - mx);
  m = mx;
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(HeroCrossEntropy a) {
  __shared__ float sm[4], ss[4];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* x = static_cast<const T*>(a.logits) + (size_t)row * a.ld;
  const float t = a.inv_temp;
  float m = -3.0e38f, s = 0.f;
  const int vec_end = ((a.ld & 7) == 0) ? (a.cols & ~7) : 0;          // 16-byte loads need aligned rows
  for (int c = threadIdx.x * 8; c < vec_end; c += 256 * 8) {
    float v[8];
    ld8<T>(x + c, v);
    float mx = v[0] * t;
#pragma unroll
    for (int k = 1; k < 8; ++k) mx = fmaxf(mx, v[k] * t);
    float e = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) e += __expf(v[k] * t - mx);
    ms_merge(m, s, mx, e);
  }
  for (int c = vec_end + threadIdx.x; c < a.cols; c += 256) ms_merge(m, s, ld1<T>(x + c) * t, 1.f);
  // wave, then block
  const float wm = wave_max(m);
  s = wave_sum(s * __expf(m - wm));
  if (lane == 0) { sm[wave] = wm; ss[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = sm[0], S = ss[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) ms_merge(M, S, sm[w], ss[w]);
    const float lse = M + __logf(S);
    const long long y = a.labels[row];
    a.lse[row] = lse;
    a.loss[row] = (y == a.ignore_index || y < 0 || y >= a.cols) ? 0.f : lse - ld1<T>(x + y) * t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(HeroCrossEntropy a) {
  const int row = blockIdx.x;
  const T* x = static_cast<const T*>(a.logits) + (size_t)row * a.ld;
  T* dx = static_cast<T*>(a.dlogits) + (size_t)row * a.ld;
  const long long y = a.labels[row];
  const bool live = !(y == a.ignore_index || y < 0 || y >= a.cols);
  const float t = a.inv_temp, lse = a.lse[row];
  const float g = live ? a.dloss[row] * t : 0.f;
  const int vec_end = ((a.ld & 7) == 0) ? (a.cols & ~7) : 0;
  for (int c = threadIdx.x * 8; c < vec_end; c += 256 * 8) {
    float v[8];
    ld8<T>(x + c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (__expf(v[k] * t - lse) - ((long long)(c + k) == y ? 1.f : 0.f)) * g;
    st8<T>(dx + c, v);
  }
  for (int c = vec_end + threadIdx.x; c < a.ld; c += 256) {
    float v = 0.f;
    if (c < a.cols) v = (__expf(ld1<T>(x + c) * t - lse) - ((long long)c == y ? 1.f : 0.f)) * g;
    st1<T>(dx + c, v);                                               // padding columns get a zero gradient
  }
}

}  // namespace
}  // namespace hero

using namespace hero;

static int check_ce(const HeroCrossEntropy* a, bool bwd) {
  HERO_REQUIRE(a && a->logits && a->labels && a->lse, "hero_cross_entropy: null pointer");
  HERO_REQUIRE(a->rows >= 0 && a->cols > 0 && a->ld >= a->cols, "hero_cross_entropy: bad dims rows=%d cols=%d ld=%d", a->rows, a->cols, a->ld);
  HERO_REQUIRE(a->dtype == HERO_F32 || a->dtype == HERO_BF16, "hero_cross_entropy: bad dtype %d", a->dtype);
  HERO_REQUIRE((((uintptr_t)a->logits) & 15) == 0, "hero_cross_entropy: logits must be 16-byte aligned");
  if (bwd) HERO_REQUIRE(a->dloss && a->dlogits && (((uintptr_t)a->dlogits) & 15) == 0, "hero_cross_entropy_bwd: dloss / dlogits required (16-byte aligned)");
  else HERO_REQUIRE(a->loss, "hero_cross_entropy_fwd: loss required");
  return HERO_OK;
}

extern "C" int hero_cross_entropy_fwd(const HeroCrossEntropy* a, hero_stream_t stream) {
  int rc = check_ce(a, false);
  if (rc) return rc;
  if (a->rows == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->dtype == HERO_BF16) hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(a->rows), dim3(256), 0, s, *a);
  else hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(a->rows), dim3(256), 0, s, *a);
  return check_launch("hero_cross_entropy_fwd");
}

extern "C" int hero_cross_entropy_bwd(const HeroCrossEntropy* a, hero_stream_t stream) {
  int rc = check_ce(a, true);
  if (rc) return rc;
  if (a->rows == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->dtype == HERO_BF16) hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(a->rows), dim3(256), 0, s, *a);
  else hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(a->rows), dim3(256), 0, s, *a);
  return check_launch("hero_cross_entropy_bwd");
}
