// GEMM lab 8 (round 3): can EIGHT compute waves that also issue their own direct-to-LDS loads run a 256 x 256 tile?
//   C[M,N] fp32 = A[K,M]^T B[K,N]   (the wgrad layout: both operands reduction-major, transposing fragment reads)
// The 192 x 192 wave-specialised kernels (4 compute + 4 loader waves) are bound by the L2 -> LDS fill of a CU
// (~76-86 GB/s): 48 KB per 64-k step = 0.56-0.63 us against 0.55 us of MFMA issue, measured 0.8 (K,K) - 1.15 (O,O)
// us per step.  A 256 x 256 tile moves 64 KB per step for 1.78 x the FLOPs (MFMA issue 0.98 us per step): the fill
// stops being the bound if the loop holds together without dedicated loader waves - 256 accumulator registers per
// wave rule them out with 4 compute waves, so all 8 waves compute a 128 x 64 sub-tile (128 accumulators) and each
// issues 1/8 of the DMA (4 x 1 KiB per 32-k stage instead of the 24 per 64-k step that sank the 4-wave self-loading
// variant of round 2).  Ring: 4 stages of 32 k (32 KB each), three in flight, one raw barrier per stage.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNO_MFMA] [-DNO_LOADS] gemm_lab8.hip -o lab8_x ; run: ./lab8_x [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int BM = 256, BN = 256, BK = 32, NS = 4;
constexpr int TM = 4, TN = 2;                       // MFMA 32 x 32 fragments per wave: 128 x 64
constexpr int A_BYTES = BK * BM * 2, B_BYTES = BK * BN * 2, STAGE = A_BYTES + B_BYTES;   // 16 + 16 KB
constexpr int LDS_BYTES = NS * STAGE;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int swz(int k) { return 4 * (k & 3); }       // 512-byte rows: see gemm_ws.hip swz_o

template <int NM, int ND>
__device__ __forceinline__ void interleave() {
  if constexpr (NM > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (ND >= 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    else if constexpr (ND == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    interleave<NM - 1, (ND >= 2 ? ND - 2 : 0)>();
  }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = N / BN;
  int wg;
  { const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
  const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
  const int nst = K / BK;
  // ---- this wave's share of the DMA: waves 0-3 the A image (4 KiB each), waves 4-7 the B image
  const bool isB = wave >= 4;
  const int wq = wave & 3;
  const int ld = isB ? N : M;
  unsigned go[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = (wq * 4 + i) * 64 + lane, row = id / 32, c = (id % 32) ^ swz(row);
    go[i] = (unsigned)row * (unsigned)ld * 2u + (c << 4);
  }
  const char* gp = reinterpret_cast<const char*>((isB ? B : A) + (isB ? n0 : m0));
  const unsigned gstep = (unsigned)BK * (unsigned)ld * 2u;
  unsigned left = (unsigned)(((size_t)K * ld - (isB ? n0 : m0)) * 2);
  unsigned fill = 0;
  int issued = 0;
  auto issue = [&]() {
    if (issued < nst) {
#ifndef NO_LOADS
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gp), 0, left, 0x00020000);
      char* buf = smem + fill + (isB ? A_BYTES : 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(buf + (wq * 4 + i) * 1024), 16, go[i], 0, 0, 0);
#endif
      gp += gstep;
      left = left > gstep ? left - gstep : 0u;
      fill += STAGE; if (fill == NS * STAGE) fill = 0;
      ++issued;
    }
  };
  // ---- compute role: 2 x 4 waves, 128 x 64 each
  const int wm = wave >> 2, wn = wave & 3;
  const int arow0 = wm * 128, brow0 = wn * 64;
  unsigned ao[TM], bo[TN];
  {
    const int p = lane & 15, gq = (lane >> 4) & 1, kg = lane >> 5;
    const int krow = kg * 8 + (p >> 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int col = arow0 + i * 32 + gq * 16 + 4 * (p & 3);
      ao[i] = krow * (BM * 2) + ((((col >> 3) ^ swz(krow)) << 4) | ((col & 7) * 2));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = brow0 + j * 32 + gq * 16 + 4 * (p & 3);
      bo[j] = A_BYTES + krow * (BN * 2) + ((((col >> 3) ^ swz(krow)) << 4) | ((col & 7) * 2));
    }
  }
  typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
  bf16x8_t a0[TM], b0[TN], a1[TM], b1[TN];
  auto ldf = [&](bf16x8_t (&a)[TM], bf16x8_t (&b)[TN], const char* st, int ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const char* q = st + ao[i] + ks * 16 * (BM * 2);
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (BM * 2)));
      a[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const char* q = st + bo[j] + ks * 16 * (BN * 2);
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (BN * 2)));
      b[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto mma = [&](const bf16x8_t (&a)[TM], const bf16x8_t (&b)[TN]) {
#ifdef NO_MFMA
    asm volatile("" ::"v"(a[0]), "v"(b[0]), "v"(a[TM - 1]), "v"(b[TN - 1]));
#else
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#endif
  };
  issue(); issue(); issue();                        // stages 0, 1, 2
  wait_vm<8>();                                     // stage 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  unsigned curo = 0;
  ldf(a0, b0, smem, 0);
  for (int u = 0; u < nst; ++u) {
    const char* cur = smem + curo;
    curo += STAGE; if (curo == NS * STAGE) curo = 0;
    const char* nxt = smem + curo;
    issue();                                        // stage u + 3 -> the slot stage u - 1 was read from before B(u - 1)
    __builtin_amdgcn_sched_barrier(0);
    ldf(a1, b1, cur, 1);
    mma(a0, b0);
    interleave<TM * TN, 2 * (TM + TN)>();
    __builtin_amdgcn_sched_barrier(0);
    wait_lds();
    if (issued - (u + 1) >= 3) wait_vm<8>(); else if (issued - (u + 1) == 2) wait_vm<4>(); else wait_vm<0>();   // stage u + 1 landed
    __builtin_amdgcn_s_barrier();                   // B(u)
    __builtin_amdgcn_sched_barrier(0);
    ldf(a0, b0, nxt, 0);
    mma(a1, b1);
    interleave<TM * TN, 2 * (TM + TN)>();
    __builtin_amdgcn_sched_barrier(0);
  }
  // register-direct fp32 store (a lab epilogue: correctness only)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int gn = n0 + brow0 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + arow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        C[(size_t)gm * N + gn] = acc[i][j][r];
      }
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 12032, M = argc > 2 ? atoi(argv[2]) : 3840, N = M;      // 15 x 15 = 225 tiles (a power-of-two row stride aliases the L2 channels)
  std::vector<uint16_t> hA((size_t)K * M), hB((size_t)K * N);
  srand(1);
  for (auto& v : hA) v = f2bf((rand() % 17 - 8) / 8.0f);
  for (auto& v : hB) v = f2bf((rand() % 13 - 6) / 8.0f);
  uint16_t *dA, *dB; float* dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int grid = (M / BM) * (N / BN);
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, N, K);
  hipDeviceSynchronize();
  printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
  std::vector<float> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int t = 0; t < 64; ++t) {
    const int m = (t * 977 + 13) % M, n = (t * 1531 + 7) % N;
    double ref = 0;
    for (int kk = 0; kk < K; ++kk) ref += (double)bf2f(hA[(size_t)kk * M + m]) * bf2f(hB[(size_t)kk * N + n]);
    worst = fmax(worst, fabs(ref - hC[(size_t)m * N + n]) / (fabs(ref) + 1.0));
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 10;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, fl = 2.0 * M * N * K;
  printf("256x256 8-wave self-loading, K = %d: %.1f us, %.0f TF/s, %.3f us per 64-k step, spot-check rel err %.2g\n", K, us, fl / us / 1e6,
         us / (K / 64.0), worst);
  return 0;
}
