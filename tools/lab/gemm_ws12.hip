// GEMM lab 12 (round 6): the one geometry DESIGN 4a-4d had not tried - a wave-specialised 256 x 192 tile at THREE
// waves per SIMD: 8 compute waves (4 x 2, each 64 x 96 = 2 x 3 MFMA 32x32 tiles: 96 accumulators, <= 168 registers) + 4 loader
// waves that only issue direct-to-LDS loads, 32-k stages (28 KB) in a five-deep ring (140 KB).  Against the product's 192 x 192
// tile it moves 12.5 % fewer DMA bytes per FLOP (56 KB per 64 k for 256 x 192 against 48 KB for 192 x 192) at the price of 25 % more
// fragment-read bytes per FLOP (every A byte is read by 2 waves, every B byte by 4) and a barrier every 12 MFMAs instead of 36.
// C[M,N] = A[M,K] B[N,K]^T, bf16 in / bf16 out, fp32 accumulate; one tile per workgroup (no persistence: the loops are the question).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNO_EPI] [-DNO_MFMA] [-DNO_LOADS] gemm_ws12.hip -o lab12_x
// Run:   ./lab12_x [M N K]...   (no args: the N = 3072 / K = 3072 shapes of the step); compare with tools/lab/gemm_ws.hip (-DTM=3 -DTN=3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#ifndef NS
#define NS 5
#endif
#ifndef GROUP
#define GROUP 8
#endif
constexpr int BM = 256, BN = 192, BK = 32;
constexpr int ROWB = BK * 2;                                  // 64-byte LDS rows: four 16-byte chunks, swizzled by (row >> 2) & 3
constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
constexpr int PA = A_BYTES / 1024 / 4, PB = B_BYTES / 1024 / 4, PW = PA + PB;   // 1-KiB pieces per loader wave per stage: 4 + 3
constexpr int TMW = 2, TNW = 3;                               // MFMA tiles per compute wave
constexpr int LDC = BN + 4;
constexpr int EPI_BYTES = (BM / 2) * LDC * 4;                 // the epilogue stages half a tile at a time
constexpr int LDS_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
static_assert(LDS_BYTES <= 160 * 1024 && (NS - 2) * PW < 64, "LDS / vmcnt budget");
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM;
  int wg;
  { const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
  const int per_group = GROUP * tiles_n, group = wg / per_group, first_m = group * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP), in_group = wg - group * per_group;
  const int pid_m = first_m + in_group % gsz, pid_n = in_group / gsz;
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = K / BK;

  f32x16_t acc[TMW][TNW];
  int arow0 = 0, brow0 = 0;

  if (wave >= 8) {
    // ------------------------------------------------------------------ loader waves (one per SIMD)
    const int w = wave - 8;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * K), 0, 0x7fffffff, 0x00020000);
    unsigned goa[PA], gob[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {                                  // a 1-KiB piece = 16 LDS rows x 4 chunks; lane -> (row, position)
      const int r = (w * PA + i) * 16 + (lane >> 2);
      goa[i] = (unsigned)(min(m0 + r, M - 1) - m0) * (unsigned)K * 2u + (((lane & 3) ^ swz(r)) << 4);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int r = (w * PB + i) * 16 + (lane >> 2);
      gob[i] = (unsigned)(min(n0 + r, N - 1) - n0) * (unsigned)K * 2u + (((lane & 3) ^ swz(r)) << 4);
    }
    unsigned fill = 0;
    auto issue = [&](int t) {
#ifndef NO_LOADS
      char* buf = smem + fill;
#pragma unroll
      for (int i = 0; i < PA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(buf + (w * PA + i) * 1024), 16, goa[i], t * ROWB, 0, 0);
#pragma unroll
      for (int i = 0; i < PB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(buf + A_BYTES + (w * PB + i) * 1024), 16, gob[i], t * ROWB, 0, 0);
#endif
      fill += STAGE;
      if (fill == NS * STAGE) fill = 0;
    };
    for (int t = 0; t < NS - 1 && t < nk; ++t) issue(t);
    if (nk >= NS - 1) wait_vm<(NS - 2) * PW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                                   // B(-1): stage 0 landed
    for (int t = 0; t < nk; ++t) {
      if (t + NS - 1 < nk) {
        issue(t + NS - 1);                                          // into the stage read during step t-1
        wait_vm<(NS - 2) * PW>();                                   // stage t+1 landed
      } else {
        wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();                                 // B(t)
    }
  } else {
    // ------------------------------------------------------------------ compute waves (two per SIMD)
    const int wm = wave >> 1, wn = wave & 1;
    arow0 = wm * TMW * 32; brow0 = wn * TNW * 32;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r = lane & 31, kg = lane >> 5;
    unsigned ao[TMW], bo[TNW];                                      // LDS byte offsets inside a stage, k-slice 0 (chunk kg); slice 1 = ^ 32
#pragma unroll
    for (int i = 0; i < TMW; ++i) { const int rr = arow0 + i * 32 + r; ao[i] = rr * ROWB + ((kg ^ swz(rr)) << 4); }
#pragma unroll
    for (int j = 0; j < TNW; ++j) { const int rr = brow0 + j * 32 + r; bo[j] = A_BYTES + rr * ROWB + ((kg ^ swz(rr)) << 4); }
    bf16x8_t a0[TMW], b0[TNW], a1[TMW], b1[TNW];
    auto ldf = [&](bf16x8_t (&a)[TMW], bf16x8_t (&b)[TNW], const char* st, int ks) {
#pragma unroll
      for (int i = 0; i < TMW; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(st + (ao[i] ^ (ks << 5)));
#pragma unroll
      for (int j = 0; j < TNW; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(st + (bo[j] ^ (ks << 5)));
    };
    auto mma = [&](const bf16x8_t (&a)[TMW], const bf16x8_t (&b)[TNW]) {
#ifndef NO_MFMA
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // swapped: lane = m row
#else
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) { acc[i][j][0] += (float)a[i][0] * (float)b[j][0]; }
#endif
    };
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_barrier();                                   // B(-1)
    ldf(a0, b0, smem, 0);
    unsigned curo = 0;
    for (int t = 0; t < nk; ++t) {
      const char* cur = smem + curo;
      curo += STAGE;
      if (curo == NS * STAGE) curo = 0;
      const char* nxt = smem + curo;
      ldf(a1, b1, cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                 // B(t): done reading `cur`, stage t+1 landed
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) ldf(a0, b0, nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  }

#ifndef NO_EPI
  // ---- epilogue (plain, two halves of 128 rows): accumulators -> LDS (fp32, padded rows) -> bf16 rows, 16 B per lane
  float* lc = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (wave < 8 && (wave >> 2) == half) {
      const int r = lane & 31, kg = lane >> 5;
      const int lrow0 = ((wave >> 1) & 1) * TMW * 32;
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4_t v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            *reinterpret_cast<f32x4_t*>(lc + (lrow0 + i * 32 + r) * LDC + brow0 + j * 32 + 8 * g + 4 * kg) = v;
          }
    }
    __syncthreads();
    {
      constexpr int C8 = BN / 8, RPI = 768 / C8;
      const int c8 = threadIdx.x % C8, r0 = threadIdx.x / C8;
      for (int row = r0; row < BM / 2; row += RPI) {
        const int gm = m0 + half * (BM / 2) + row;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(lc + row * LDC + c8 * 8);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(lc + row * LDC + c8 * 8 + 4);
        uint4 o;
        o.x = (__float_as_uint(v0[0]) >> 16) | (__float_as_uint(v0[1]) & 0xffff0000u);
        o.y = (__float_as_uint(v0[2]) >> 16) | (__float_as_uint(v0[3]) & 0xffff0000u);
        o.z = (__float_as_uint(v1[0]) >> 16) | (__float_as_uint(v1[1]) & 0xffff0000u);
        o.w = (__float_as_uint(v1[2]) >> 16) | (__float_as_uint(v1[3]) & 0xffff0000u);
        if (gm < M) *reinterpret_cast<uint4*>(C + (size_t)gm * N + n0 + c8 * 8) = o;
      }
    }
  }
#else
  if (wave < 8) {                                                   // every accumulator stays live (a single-element test lets the compiler delete MFMAs)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    if (sum == 123.456f) C[0] = 1;
  }
#endif
}

// naive reference (fp32 accumulate, truncating bf16 store like the lab kernel)
__global__ void ref_k(const uint16_t* A, const uint16_t* B, uint16_t* C, int M, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int kk = 0; kk < K; ++kk)
    s += __uint_as_float((uint32_t)A[(size_t)m * K + kk] << 16) * __uint_as_float((uint32_t)B[(size_t)n * K + kk] << 16);
  C[(size_t)m * N + n] = (uint16_t)(__float_as_uint(s) >> 16);
}

static uint16_t rnd_bf16() {   // uniform [-1, 1)
  const float f = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
  union { float x; uint32_t u; } c; c.x = f;
  return (uint16_t)(c.u >> 16);
}

static void run(int M, int N, int K, bool verify) {
  uint16_t *A, *B, *C, *R;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&R, (size_t)M * N * 2);
  std::vector<uint16_t> h((size_t)M * K);
  for (auto& x : h) x = rnd_bf16();
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  h.resize((size_t)N * K); for (auto& x : h) x = rnd_bf16();
  hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemset(C, 0xff, (size_t)M * N * 2);
  const int lds = LDS_BYTES;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = ((M + BM - 1) / BM) * (N / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // past the clock ramp (tools/lab/warm_probe.py: the first ~70 ms after idle run up to 20 % slower)
  for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(768), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e0);
  const int n = 40;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(768), lds, 0, A, B, C, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / n;
  double worst = -1; long bad = 0;
#ifndef NO_EPI
  if (verify) {
    hipLaunchKernelGGL(ref_k, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, R, M, N, K);
    std::vector<uint16_t> hc((size_t)M * N), hr((size_t)M * N);
    hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), R, hr.size() * 2, hipMemcpyDeviceToHost);
    auto f = [](uint16_t v) { union { uint32_t u; float x; } c; c.u = (uint32_t)v << 16; return c.x; };
    worst = 0;
    for (size_t i = 0; i < hc.size(); ++i) {
      const double d = fabs(f(hc[i]) - f(hr[i])), tol = 0.02 * fabs(f(hr[i])) + 0.05;
      if (!(d <= tol)) ++bad;
      if (d > worst) worst = d;
    }
  }
#endif
  printf("ws12 256x192x32 NS=%d lds=%dK grid=%d (%.2f rounds) M=%d N=%d K=%d  %.1f us  %.1f TF/s  maxabs %.3g bad %ld (%s)\n", NS, lds >> 10, grid,
         grid / 256.0, M, N, K, us, 2.0 * M * N * K / us / 1e6, worst, bad, hipGetErrorString(hipGetLastError()));
  hipFree(A); hipFree(B); hipFree(C); hipFree(R);
}

int main(int argc, char** argv) {
  if (argc >= 4) {
    for (int i = 1; i + 2 < argc; i += 3) run(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), true);
    return 0;
  }
  const int shapes[][3] = {{12000, 3072, 768}, {12000, 768, 3072}, {12000, 2304, 768}, {12032, 3072, 3072}};
  for (auto& s : shapes) run(s[0], s[1], s[2], true);
  return 0;
}
