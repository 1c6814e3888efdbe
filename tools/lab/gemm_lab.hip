// Stand-alone GEMM lab: C[M,N] = A[M,K] * B[N,K]^T, bf16, 128x128x64 tile, 4 waves, register staging.
// Ablation flags: -DNO_LOADS (skip global loads after the first tile), -DNO_MFMA, -DNO_STORE (skip LDS writes),
// -DNO_EPI (skip epilogue), -DBLOCKS_PER_CU hint via launch bounds, -DGLDS (direct-to-LDS loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#ifndef WAVES_EU
#define WAVES_EU 2
#endif
__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WAVES_EU)))
void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = N / 128, tiles_m = M / 128;
#ifdef REMAP
  int wg;
  { const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
  const int per_group = 8 * tiles_n, group = wg / per_group, first_m = group * 8;
  const int gsz = min(tiles_m - first_m, 8), in_group = wg - group * per_group;
  const int pid_m = first_m + in_group % gsz, pid_n = in_group / gsz;
#else
  const int pid_m = blockIdx.x / tiles_n, pid_n = blockIdx.x % tiles_n;
#endif
  const int m0 = pid_m * 128, n0 = pid_n * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  f32x16_t acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int goffa[4], goffb[4], loff[4];
  for (int i = 0; i < 4; ++i) {
    const int q = threadIdx.x + 256 * i, r = q >> 3, c = q & 7;
    goffa[i] = r * K + c * 8; goffb[i] = r * K + c * 8; loff[i] = r * 128 + ((c ^ swz(r)) << 4);
  }
  const uint16_t* pa = A + (size_t)m0 * K;
  const uint16_t* pb = B + (size_t)n0 * K;
  uint4 va[4], vb[4];
  const int nk = K / 64;
#ifdef GLDS
  // direct-to-LDS: wave w of the block fills rows [w*32 + i*8 .. ) : lane -> (row = lane>>3, chunk' = lane&7), source chunk = chunk' ^ swz(row)
  auto glds_tile = [&](char* buf, const uint16_t* ga, const uint16_t* gb) {
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 32 + i * 8 + (lane >> 3), cp = lane & 7, c = cp ^ swz(r);
      __builtin_amdgcn_global_load_lds(ga + (size_t)r * K + c * 8, (__attribute__((address_space(3))) void*)(buf + (wave * 32 + i * 8) * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gb + (size_t)r * K + c * 8, (__attribute__((address_space(3))) void*)(buf + 16384 + (wave * 32 + i * 8) * 128), 16, 0, 0);
    }
  };
  glds_tile(smem, pa, pb);
  __syncthreads();
#else
  for (int i = 0; i < 4; ++i) { va[i] = *(const uint4*)(pa + goffa[i]); vb[i] = *(const uint4*)(pb + goffb[i]); }
  for (int i = 0; i < 4; ++i) { *(uint4*)(smem + loff[i]) = va[i]; *(uint4*)(smem + 16384 + loff[i]) = vb[i]; }
  __syncthreads();
#endif
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 32768;
    char* nxt = smem + ((kt + 1) & 1) * 32768;
    const bool more = kt + 1 < nk;
#ifndef NO_LOADS
    if (more) {
      pa += 64; pb += 64;
#ifdef GLDS
      glds_tile(nxt, pa, pb);
#else
      for (int i = 0; i < 4; ++i) { va[i] = *(const uint4*)(pa + goffa[i]); vb[i] = *(const uint4*)(pb + goffb[i]); }
#endif
    }
#endif
#ifndef NO_MFMA
    {
      const int r = lane & 31, kg = lane >> 5;
      for (int ks = 0; ks < 4; ++ks) {
        const int kc = ks * 2 + kg;
        bf16x8_t a[2], b[2];
        for (int i = 0; i < 2; ++i) {
          const int ra = wm * 64 + i * 32 + r; a[i] = *(const bf16x8_t*)(cur + ra * 128 + ((kc ^ swz(ra)) << 4));
          const int rb = wn * 64 + i * 32 + r; b[i] = *(const bf16x8_t*)(cur + 16384 + rb * 128 + ((kc ^ swz(rb)) << 4));
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
#endif
#if !defined(NO_STORE) && !defined(GLDS)
    if (more) for (int i = 0; i < 4; ++i) { *(uint4*)(nxt + loff[i]) = va[i]; *(uint4*)(nxt + 16384 + loff[i]) = vb[i]; }
#else
    asm volatile("" :: "v"(va[0].x), "v"(vb[0].x), "v"(va[3].w), "v"(vb[3].w));
#endif
    __syncthreads();
  }
#if defined(EPI_LDS)
  {
    float* lc = (float*)smem;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
      const int col = wn * 64 + j * 32 + (lane & 31);
      for (int r = 0; r < 16; ++r) lc[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128 + col] = acc[i][j][r];
    }
    __syncthreads();
    const int c4 = (threadIdx.x & 31) * 4;
    for (int it = 0; it < 16; ++it) {
      const int row = (threadIdx.x >> 5) + it * 8;
      const float4 v = *(const float4*)(lc + row * 128 + c4);
      uint2 o; o.x = (__float_as_uint(v.x) >> 16) | (__float_as_uint(v.y) & 0xffff0000u); o.y = (__float_as_uint(v.z) >> 16) | (__float_as_uint(v.w) & 0xffff0000u);
      *(uint2*)(C + (size_t)(m0 + row) * N + n0 + c4) = o;
    }
  }
#elif !defined(NO_EPI)
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
    const int gn = n0 + wn * 64 + j * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      C[(size_t)gm * N + gn] = (uint16_t)(__float_as_uint(acc[i][j][r]) >> 16);
    }
  }
#else
  if (acc[0][0][0] == 123.456f) C[0] = 1;
#endif
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 11520, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  uint16_t *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& x : h) x = 0x3c00 + (rand() & 0xff);   // ~[0.0078..]
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  h.resize((size_t)N * K); for (auto& x : h) x = 0xbc00 + (rand() & 0x3ff);
  hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int grid = (M / 128) * (N / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, 0, A, B, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / n;
  printf("%s M=%d N=%d K=%d  %.1f us  %.1f TF/s  (%s)\n", argv[0], M, N, K, us, 2.0 * M * N * K / us / 1e6, hipGetErrorString(hipGetLastError()));
  return 0;
}
