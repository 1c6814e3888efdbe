// What MFMA rate does the chip sustain?  8 waves per CU (2 per SIMD) or 4 (1 per SIMD), each issuing independent
// v_mfma_f32_32x32x16_bf16 back to back (9 accumulators, like the 96 x 96 wave tile of gemm_ws.hip), no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
  f32x16_t acc[9];
  for (int i = 0; i < 9; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 9; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* d; long long* c; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 8);
  for (int threads : {256, 512}) {
    const int iters = 2000;
    for (int w = 0; w < 300; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 2000, c);   // ~0.3 s: past the clock ramp
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double us_per_36 = ms * 1e3 / iters;          // per 36 MFMAs of one wave (= one 64-k step of a 96 x 96 wave tile)
    const double flops = 2.0 * 32 * 32 * 16 * 36.0 * iters * (threads / 64) * 256;
    printf("%d waves/CU: %.3f us per 36 MFMAs per wave, %.0f TF/s, %.0f shader cycles per 36 MFMAs (s_memtime), clock ~%.2f GHz\n",
           threads / 64, us_per_36, flops / (ms * 1e-3) / 1e12, (double)cy / iters, (double)cy / iters / us_per_36 / 1e3);
  }
  return 0;
}
