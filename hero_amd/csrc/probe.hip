// Box probes for bench.py: what THIS box's matrix pipes and HBM deliver right now, measured the same way on every box, so
// that a bench line can be normalised across the pool (the same tree ran 6.37-7.49 ms per step on different boxes in round
// 4; the MFMA clocks move, the memory does not).  Unlike the rest of the library these two entries time themselves with
// HIP events and therefore SYNCHRONISE the stream they are given (like hero_prof_read); never call them under capture.
#include "common.h"

namespace hero {
namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// 8 waves per CU (2 per SIMD, like the persistent GEMMs), each issuing 36 independent v_mfma_f32_32x32x16_bf16 per
// iteration (the 9 accumulators x 4 k-slices of one 64-k step of a 96 x 96 wave tile), no memory traffic.
__global__ void __launch_bounds__(512) mfma_peak_kernel(float* out, int iters, long long* cyc) {
  f32x16_t acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) s += acc[i][0];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// streaming copy / read: 16 bytes per lane, grid-stride, every load of a thread's four in flight together
__global__ void __launch_bounds__(256) stream_copy_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4_t a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const f32x4_t c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride);
    __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ void __launch_bounds__(256) stream_read_kernel(const f32x4_t* __restrict__ src, float* __restrict__ sink, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4_t a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const f32x4_t c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc += (a + b) + (c + d);
  }
  for (; i < n4; i += stride) acc += __builtin_nontemporal_load(src + i);
  const float t = acc.x + acc.y + acc.z + acc.w;
  if (t == 123.456f) sink[0] = t;            // never true for the zero / finite contents the caller provides; keeps the loads
}

// the same copy with ordinary (cached) accesses: which of the two streams faster differs with the part's cache policy; the probe
// reports the better one
__global__ void __launch_bounds__(256) stream_copy_plain_kernel(const f32x4_t* __restrict__ src, f32x4_t* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {
    f32x4_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[i + k * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[i + k * stride] = v[k];
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

struct Ev {
  hipEvent_t a = nullptr, b = nullptr;
  bool ok() { return hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
  ~Ev() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

}  // namespace
}  // namespace hero

using namespace hero;

extern "C" int hero_probe_mfma(void* scratch, size_t scratch_bytes, double* tflops, double* ghz, hero_stream_t stream) {
  HERO_REQUIRE(scratch && tflops && ghz, "hero_probe_mfma: null pointer");
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  HERO_REQUIRE(scratch_bytes >= (size_t)cus * 512 * 4 + 16, "hero_probe_mfma: scratch must hold %zu bytes", (size_t)cus * 512 * 4 + 16);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* out = static_cast<float*>(scratch);
  long long* cyc = reinterpret_cast<long long*>(static_cast<char*>(scratch) + (((size_t)cus * 512 * 4 + 7) & ~(size_t)7));
  Ev ev;
  if (!ev.ok()) { set_error("hero_probe_mfma: hipEventCreate failed"); return HERO_ERR_LAUNCH; }
  const int iters = 2000;                                               // ~1 ms per launch
  for (int w = 0; w < 80; ++w) hipLaunchKernelGGL(mfma_peak_kernel, dim3(cus), dim3(512), 0, s, out, iters, cyc);   // past the clock ramp (~70 ms)
  double best_ms = 1e30;
  long long best_cyc = 0;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(ev.a, s);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(cus), dim3(512), 0, s, out, iters, cyc);
    (void)hipEventRecord(ev.b, s);
    if (hipEventSynchronize(ev.b) != hipSuccess) { set_error("hero_probe_mfma: synchronise failed"); return HERO_ERR_LAUNCH; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ev.a, ev.b);
    long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (ms < best_ms) { best_ms = ms; best_cyc = c; }
  }
  const double flops = 2.0 * 32 * 32 * 16 * 36.0 * iters * 8.0 * cus;
  *tflops = flops / (best_ms * 1e-3) / 1e12;
  // sustained MATRIX clock from the MFMA occupancy: a 32x32x16 bf16 MFMA holds a SIMD's matrix pipe for 8 passes x 4 cycles and
  // the two waves of a SIMD alternate, so an iteration is 2 x 36 x 32 pipe cycles.  (s_memtime / __builtin_readcyclecounter is
  // NOT the shader clock on this part: it read 1.21 GHz while the pipes demonstrably ran at 2.35.)
  *ghz = 2.0 * 36.0 * 32.0 * iters / (best_ms * 1e-3) / 1e9;
  (void)best_cyc;
  return check_launch("hero_probe_mfma");
}

extern "C" int hero_probe_hbm(const void* src, void* dst, size_t bytes, double* copy_gbps, double* read_gbps, hero_stream_t stream) {
  HERO_REQUIRE(src && dst && copy_gbps && read_gbps, "hero_probe_hbm: null pointer");
  HERO_REQUIRE(bytes >= ((size_t)1 << 28) && bytes % 16 == 0, "hero_probe_hbm: buffers of >= 256 MiB (beyond the Infinity Cache), multiple of 16 bytes");
  HERO_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "hero_probe_hbm: buffers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  Ev ev;
  if (!ev.ok()) { set_error("hero_probe_hbm: hipEventCreate failed"); return HERO_ERR_LAUNCH; }
  const size_t n4 = bytes / 16;
  const int grid = cus * 8;
  double best[3] = {1e30, 1e30, 1e30};
  for (int kind = 0; kind < 3; ++kind)
    for (int rep = 0; rep < 4; ++rep) {                                  // rep 0 is the warm-up
      (void)hipEventRecord(ev.a, s);
      if (kind == 0) hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, s, static_cast<const f32x4_t*>(src), static_cast<f32x4_t*>(dst), n4);
      else if (kind == 2) hipLaunchKernelGGL(stream_copy_plain_kernel, dim3(grid), dim3(256), 0, s, static_cast<const f32x4_t*>(src), static_cast<f32x4_t*>(dst), n4);
      else hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, s, static_cast<const f32x4_t*>(src), static_cast<float*>(dst), n4);
      (void)hipEventRecord(ev.b, s);
      if (hipEventSynchronize(ev.b) != hipSuccess) { set_error("hero_probe_hbm: synchronise failed"); return HERO_ERR_LAUNCH; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev.a, ev.b);
      if (rep > 0 && ms < best[kind]) best[kind] = ms;
    }
  *copy_gbps = 2.0 * (double)bytes / ((best[0] < best[2] ? best[0] : best[2]) * 1e-3) / 1e9;      // bytes read + bytes written, the better of the two copies
  *read_gbps = (double)bytes / (best[1] * 1e-3) / 1e9;
  return check_launch("hero_probe_hbm");
}
