// GEMM lab 3: C[M,N] = A[M,K] B[N,K]^T bf16; tile (64*WM) x (64*WN) x 64, WM*WN waves of 64x64 each,
// direct-to-LDS staging into an S-stage ring with COUNTED vmcnt (prefetch distance S-1 tiles).
// Build: hipcc --offload-arch=gfx950 -O3 -DWM=4 -DWN=2 -DS=3 [-DNO_EPI] [-DNO_MFMA] gemm_lab3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#ifndef WM
#define WM 2
#endif
#ifndef WN
#define WN 2
#endif
#ifndef S
#define S 2
#endif
#ifndef WEU
#define WEU 2
#endif
constexpr int NW = WM * WN, NT = 64 * NW, BM = 64 * WM, BN = 64 * WN;
constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW, PW = PA + PB;   // glds instructions per wave per tile
static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");
__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, WEU)))
void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM;
  int wg;
  { const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
  const int per_group = 8 * tiles_n, group = wg / per_group, first_m = group * 8;
  const int gsz = min(tiles_m - first_m, 8), in_group = wg - group * per_group;
  const int pid_m = first_m + in_group % gsz, pid_n = in_group / gsz;
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / WN, wn = wave % WN;
  f32x16_t acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // this wave's glds pieces: A rows [wave*PA*8 + i*8, +8), B rows [wave*PB*8 + i*8, +8)
  int goa[PA], gob[PB];
  for (int i = 0; i < PA; ++i) { const int r = wave * PA * 8 + i * 8 + (lane >> 3); goa[i] = min(m0 + r, M - 1) * K + (((lane & 7) ^ swz(r)) << 3); }
  for (int i = 0; i < PB; ++i) { const int r = wave * PB * 8 + i * 8 + (lane >> 3); gob[i] = (n0 + r) * K + (((lane & 7) ^ swz(r)) << 3); }
  auto issue = [&](int kt) {
    char* buf = smem + (kt % S) * STAGE;
    const uint16_t* ga = A + kt * 64;
    const uint16_t* gb = B + kt * 64;
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_global_load_lds(ga + goa[i], (__attribute__((address_space(3))) void*)(buf + (wave * PA * 8 + i * 8) * 128), 16, 0, 0);
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_global_load_lds(gb + gob[i], (__attribute__((address_space(3))) void*)(buf + A_BYTES + (wave * PB * 8 + i * 8) * 128), 16, 0, 0);
  };
  const int nk = K / 64;
  for (int t = 0; t < S - 1 && t < nk; ++t) issue(t);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed: tiles kt+1 .. kt+S-2 may still be in flight
    if (kt + S - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // everyone's pieces of tile kt landed; everyone done with tile kt-1
    if (kt + S - 1 < nk) issue(kt + S - 1);           // overwrites the stage of tile kt-1
#ifndef NO_MFMA
    const char* cur = smem + (kt % S) * STAGE;
    const int r = lane & 31, kg = lane >> 5;
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + kg;
      bf16x8_t a[2], b[2];
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + r; a[i] = *(const bf16x8_t*)(cur + ra * 128 + ((kc ^ swz(ra)) << 4));
        const int rb = wn * 64 + i * 32 + r; b[i] = *(const bf16x8_t*)(cur + A_BYTES + rb * 128 + ((kc ^ swz(rb)) << 4));
      }
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#endif
  }
#ifndef NO_EPI
  __syncthreads();
  {  // LDS-staged epilogue, 64 rows per wave-row group: each wave stages its own 64x64 and writes it
    float* lc = (float*)smem + wave * 64 * 68;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) lc[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 68 + j * 32 + (lane & 31)] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + (lane >> 4), c4 = (lane & 15) * 4;
      const float4 v = *(const float4*)(lc + row * 68 + c4);
      const int gm = m0 + wm * 64 + row;
      uint2 o; o.x = (__float_as_uint(v.x) >> 16) | (__float_as_uint(v.y) & 0xffff0000u); o.y = (__float_as_uint(v.z) >> 16) | (__float_as_uint(v.w) & 0xffff0000u);
      if (gm < M) *(uint2*)(C + (size_t)gm * N + n0 + wn * 64 + c4) = o;
    }
  }
#else
  if (acc[0][0][0] == 123.456f) C[0] = 1;
#endif
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
  int lds = S * STAGE; if (lds < NW * 64 * 68 * 4) lds = NW * 64 * 68 * 4;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = ((M + BM - 1) / BM) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / n;
  // spot check against a host reference on a few elements
  std::vector<uint16_t> hc((size_t)M * N), ha((size_t)M * K), hb((size_t)N * K);
  hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), A, ha.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), B, hb.size() * 2, hipMemcpyDeviceToHost);
  auto f = [](uint16_t v) { union { uint32_t u; float x; } c; c.u = (uint32_t)v << 16; return c.x; };
  double worst = 0;
  for (int t = 0; t < 64; ++t) {
    const int m = (t * 7919) % M, nn = (t * 104729) % N;
    double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)f(ha[(size_t)m * K + kk]) * f(hb[(size_t)nn * K + kk]);
    const double err = fabs(ref - f(hc[(size_t)m * N + nn])) / (fabs(ref) + 1e-3);
    if (err > worst) worst = err;
  }
  printf("%s WM=%d WN=%d S=%d lds=%dK grid=%d M=%d N=%d K=%d  %.1f us  %.1f TF/s  relerr %.3g (%s)\n", argv[0], WM, WN, S, lds >> 10, grid, M, N, K, us,
         2.0 * M * N * K / us / 1e6, worst, hipGetErrorString(hipGetLastError()));
  return 0;
}
