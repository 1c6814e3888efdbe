// Lab: glds GEMM 128x128 tile with BK=32 (16 KB/stage -> 32 KB LDS, up to 4-5 workgroups per CU) vs BK=64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#ifndef BK
#define BK 32
#endif
#ifndef WEU
#define WEU 4
#endif
constexpr int ROWB = BK * 2;            // bytes per tile row
constexpr int CH = ROWB / 16;           // 16-B chunks per row (4 or 8)
__device__ __forceinline__ int swz(int row) { return CH == 8 ? ((row ^ (row >> 3)) & 7) : ((row >> 2) & 3); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WEU)))
void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = N / 128, tiles_m = M / 128;
  int wg;
  { const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
  const int per_group = 8 * tiles_n, group = wg / per_group, first_m = group * 8;
  const int gsz = min(tiles_m - first_m, 8), in_group = wg - group * per_group;
  const int pid_m = first_m + in_group % gsz, pid_n = in_group / gsz;
  const int m0 = pid_m * 128, n0 = pid_n * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  f32x16_t acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int TILE = 128 * ROWB;              // bytes per operand tile
  constexpr int RPI = 1024 / ROWB;              // rows per wave instruction (8 or 16)
  constexpr int NI = 128 / RPI / 4;             // instructions per wave per operand
  const uint16_t* pa = A + (size_t)m0 * K;
  const uint16_t* pb = B + (size_t)n0 * K;
  auto glds_tile = [&](char* buf, const uint16_t* ga, const uint16_t* gb) {
    for (int i = 0; i < NI; ++i) {
      const int rbase = wave * (128 / 4) + i * RPI;
      const int r = rbase + lane / CH, cp = lane % CH, c = cp ^ swz(r);
      __builtin_amdgcn_global_load_lds(ga + (size_t)r * K + c * 8, (__attribute__((address_space(3))) void*)(buf + rbase * ROWB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gb + (size_t)r * K + c * 8, (__attribute__((address_space(3))) void*)(buf + TILE + rbase * ROWB), 16, 0, 0);
    }
  };
  const int nk = K / BK;
  glds_tile(smem, pa, pb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * TILE;
    char* nxt = smem + ((kt + 1) & 1) * 2 * TILE;
    if (kt + 1 < nk) { pa += BK; pb += BK; glds_tile(nxt, pa, pb); }
    const int r = lane & 31, kg = lane >> 5;
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int kc = ks * 2 + kg;
      bf16x8_t a[2], b[2];
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + r; a[i] = *(const bf16x8_t*)(cur + ra * ROWB + ((kc ^ swz(ra)) << 4));
        const int rb = wn * 64 + i * 32 + r; b[i] = *(const bf16x8_t*)(cur + TILE + rb * ROWB + ((kc ^ swz(rb)) << 4));
      }
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
    const int gn = n0 + wn * 64 + j * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      C[(size_t)gm * N + gn] = (uint16_t)(__float_as_uint(acc[i][j][r]) >> 16);
    }
  }
}

int main(int argc, char** argv) {
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  uint16_t *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& x : h) x = 0x3c00 + (rand() & 0xff);
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  h.resize((size_t)N * K); for (auto& x : h) x = 0xbc00 + (rand() & 0x3ff);
  hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int lds = 4 * 128 * ROWB;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = (M / 128) * (N / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / n;
  // spot check vs host for a few entries
  std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K), hc((size_t)M * N);
  hipMemcpy(ha.data(), A, ha.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), B, hb.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
  auto f = [](uint16_t v) { union { uint32_t u; float x; } c; c.u = (uint32_t)v << 16; return c.x; };
  double maxrel = 0;
  for (int t = 0; t < 64; ++t) {
    const int m = (t * 7919) % M, nn = (t * 104729) % N;
    double s = 0; for (int kk = 0; kk < K; ++kk) s += (double)f(ha[(size_t)m * K + kk]) * f(hb[(size_t)nn * K + kk]);
    const double got = f(hc[(size_t)m * N + nn]);
    maxrel = fmax(maxrel, fabs(got - s) / (fabs(s) + 1e-6));
  }
  printf("%s BK=%d WEU=%d M=%d N=%d K=%d  %.1f us  %.1f TF/s  maxrel %.3g (%s)\n", argv[0], BK, WEU, M, N, K, us, 2.0 * M * N * K / us / 1e6, maxrel, hipGetErrorString(hipGetLastError()));
  return 0;
}
