// GEMM lab 4: persistent 128x128x64 workgroups (4 waves, 2-stage direct-to-LDS ring) that issue the
// first K tile of their NEXT output tile before running the epilogue of the current one, so the
// epilogue's stores and the next tile's load latency overlap.  Epilogue stages through ONE stage buffer
// (two 64-row passes), the other one receives the prefetch.
// Build: hipcc --offload-arch=gfx950 -O3 [-DPERSIST=0|1] [-DEARLY=0|1] gemm_lab4.hip -o lab4_x ; ./lab4_x M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#ifndef PERSIST
#define PERSIST 1
#endif
#ifndef EARLY
#define EARLY 1
#endif
constexpr int BM = 128, BN = 128, STAGE = 32768;
__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

__device__ __forceinline__ void tile_of(int wg, int nwg, int tiles_m, int tiles_n, int& m0, int& n0) {
  const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int per_group = 8 * tiles_n, group = t / per_group, first_m = group * 8;
  const int gsz = min(tiles_m - first_m, 8), in_group = t - group * per_group;
  m0 = (first_m + in_group % gsz) * BM;
  n0 = (in_group / gsz) * BN;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int nk = K / 64;
  int goa[4], gob[4];
  auto setup = [&](int m0, int n0) {
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 32 + i * 8 + (lane >> 3);
      goa[i] = min(m0 + r, M - 1) * K + (((lane & 7) ^ swz(r)) << 3);
      gob[i] = (n0 + r) * K + (((lane & 7) ^ swz(r)) << 3);
    }
  };
  auto issue = [&](int kt, char* buf) {
    const uint16_t* ga = A + kt * 64;
    const uint16_t* gb = B + kt * 64;
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds(ga + goa[i], (__attribute__((address_space(3))) void*)(buf + (wave * 32 + i * 8) * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gb + gob[i], (__attribute__((address_space(3))) void*)(buf + 16384 + (wave * 32 + i * 8) * 128), 16, 0, 0);
    }
  };
  const int step = PERSIST ? gridDim.x : ntiles;
  int tile = blockIdx.x;
  int m0, n0;
  tile_of(tile, ntiles, tiles_m, tiles_n, m0, n0);
  setup(m0, n0);
  issue(0, smem);                                      // nk is even: tile's K tile kt lives in buffer kt & 1
  while (true) {
    f32x16_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      char* cur = smem + (kt & 1) * STAGE;
      if (kt + 1 < nk) issue(kt + 1, smem + ((kt + 1) & 1) * STAGE);
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // the last K tile sat in buffer 1 (nk even) and everyone is past the barrier: both buffers are free.
    const int em0 = m0, en0 = n0;
    const int next = tile + step;
    const bool more = next < ntiles;
    if (more) {
      tile_of(next, ntiles, tiles_m, tiles_n, m0, n0);
      setup(m0, n0);
      if (EARLY) issue(0, smem);                       // prefetch the next tile's first K tile into buffer 0
    }
    // epilogue through buffer 1 only: two passes of 64 rows (wm selects the pass)
    float* lc = (float*)(smem + STAGE);
    for (int pass = 0; pass < 2; ++pass) {
      if (pass) __syncthreads();
      if (wm == pass)
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
          const int col = wn * 64 + j * 32 + (lane & 31);
          for (int r = 0; r < 16; ++r) lc[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128 + col] = acc[i][j][r];
        }
      __syncthreads();
      const int c4 = (threadIdx.x & 31) * 4;
      for (int it = 0; it < 8; ++it) {
        const int row = (threadIdx.x >> 5) + it * 8;
        const float4 v = *(const float4*)(lc + row * 128 + c4);
        const int gm = em0 + pass * 64 + row;
        uint2 o; o.x = (__float_as_uint(v.x) >> 16) | (__float_as_uint(v.y) & 0xffff0000u); o.y = (__float_as_uint(v.z) >> 16) | (__float_as_uint(v.w) & 0xffff0000u);
        if (gm < M) *(uint2*)(C + (size_t)gm * N + en0 + c4) = o;
      }
    }
    if (!more) break;
    tile = next;
    if (!EARLY) issue(0, smem);
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 12000, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  uint16_t *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& x : h) x = 0x3c00 + (rand() & 0xff);
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  h.resize((size_t)N * K); for (auto& x : h) x = 0xbc00 + (rand() & 0x3ff);
  hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int lds = 65536;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int ntiles = ((M + BM - 1) / BM) * (N / BN);
  const int grid = PERSIST ? (ntiles < 512 ? ntiles : 512) : ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / n;
  std::vector<uint16_t> hc((size_t)M * N), ha((size_t)M * K), hb((size_t)N * K);
  hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), A, ha.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), B, hb.size() * 2, hipMemcpyDeviceToHost);
  auto f = [](uint16_t v) { union { uint32_t u; float x; } c; c.u = (uint32_t)v << 16; return c.x; };
  double worst = 0;
  for (int t = 0; t < 256; ++t) {
    const int m = (int)(((long long)t * 7919 + 13) % M), nn = (int)(((long long)t * 104729 + 7) % N);
    double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)f(ha[(size_t)m * K + kk]) * f(hb[(size_t)nn * K + kk]);
    const double err = fabs(ref - f(hc[(size_t)m * N + nn])) / (fabs(ref) + 1e-3);
    if (err > worst) worst = err;
  }
  printf("PERSIST=%d EARLY=%d grid=%d M=%d N=%d K=%d  %.1f us  %.1f TF/s  relerr %.3g (%s)\n", PERSIST, EARLY, grid, M, N, K, us,
         2.0 * M * N * K / us / 1e6, worst, hipGetErrorString(hipGetLastError()));
  return 0;
}
