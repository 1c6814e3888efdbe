// MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   C[M,N] = op(A) * op(B)   tile 128x128, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles
//   bf16: v_mfma_f32_32x32x16_bf16, BK = 64 ; f32: v_mfma_f32_32x32x2_f32 (exact f32), BK = 32
//
// LDS image of an operand tile is always [128 rows][128 bytes] with the eight 16-byte chunks of a
// row XOR-swizzled by swz(row) = (row ^ row>>3) & 7, whatever the operand's layout in HBM:
//   HERO_LAYOUT_K (reduction contiguous)  global 16 B -> LDS 16 B
//   HERO_LAYOUT_O (outer dim contiguous)  4 reduction rows x 16 B are transposed in registers
//                                         and written as 8 B (bf16) / 16 B (f32) pieces
// so forward (x W^T), dgrad (dY W) and wgrad (dY^T X) run on the same main loop without any
// transposed copies of weights or activations in HBM.
// Staging is register-staged and split (issue loads for tile t+1, compute tile t, write LDS),
// double-buffered, one barrier per K tile. The epilogue goes through LDS so that bias / residual /
// aux / output traffic is 8-16 B per lane and coalesced along N.
#include "common.h"
#include <vector>

namespace hero {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int BM = 128, BN = 128, NT = 256;

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

template <typename T> struct Tr;
template <> struct Tr<bf16_t> { static constexpr int BK = 64, EPC = 8; };
template <> struct Tr<float>  { static constexpr int BK = 32, EPC = 4; };

// ---------------------------------------------------------------------------------------------
// global -> register -> LDS staging of one 128 x BK operand tile
// ---------------------------------------------------------------------------------------------
template <typename T, int LAY> struct Stage;

template <typename T> struct Stage<T, HERO_LAYOUT_K> {
  uint4 v[4];
  __device__ __forceinline__ void load(const T* __restrict__ base, int ld, int row0, int nrows, int k0, int kend) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = threadIdx.x + NT * i;
      const int r = q >> 3, c = q & 7;
      const int gr = row0 + r, gk = k0 + c * Tr<T>::EPC;
      if (gr < nrows && gk < kend)
        v[i] = *reinterpret_cast<const uint4*>(base + (size_t)gr * ld + gk);
      else
        v[i] = make_uint4(0, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = threadIdx.x + NT * i;
      const int r = q >> 3, c = q & 7;
      *reinterpret_cast<uint4*>(lds + r * 128 + ((c ^ swz(r)) << 4)) = v[i];
    }
  }
};

template <> struct Stage<bf16_t, HERO_LAYOUT_O> {
  uint4 v[4];  // 4 reduction rows x 8 outer elements
  __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int ld, int row0, int nrows, int k0, int kend) {
    const int og = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int go = row0 + og * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gk = k0 + rg * 4 + j;
      if (gk < kend && go < nrows)
        v[j] = *reinterpret_cast<const uint4*>(base + (size_t)gk * ld + go);
      else
        v[j] = make_uint4(0, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
    const int og = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const uint32_t w0[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
    const uint32_t w1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
    const uint32_t w2[4] = {v[2].x, v[2].y, v[2].z, v[2].w};
    const uint32_t w3[4] = {v[3].x, v[3].y, v[3].z, v[3].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int w = e >> 1;
      uint2 o;
      if (e & 1) {
        o.x = (w0[w] >> 16) | (w1[w] & 0xffff0000u);
        o.y = (w2[w] >> 16) | (w3[w] & 0xffff0000u);
      } else {
        o.x = (w0[w] & 0xffffu) | (w1[w] << 16);
        o.y = (w2[w] & 0xffffu) | (w3[w] << 16);
      }
      const int r = og * 8 + e;
      *reinterpret_cast<uint2*>(lds + r * 128 + (((rg >> 1) ^ swz(r)) << 4) + ((rg & 1) << 3)) = o;
    }
  }
};

template <> struct Stage<float, HERO_LAYOUT_O> {
  float4 v[4];  // 4 reduction rows x 4 outer elements
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int row0, int nrows, int k0, int kend) {
    const int og = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int go = row0 + og * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gk = k0 + rg * 4 + j;
      if (gk < kend && go < nrows)
        v[j] = *reinterpret_cast<const float4*>(base + (size_t)gk * ld + go);
      else
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
    const int og = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const float c0[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
    const float c1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
    const float c2[4] = {v[2].x, v[2].y, v[2].z, v[2].w};
    const float c3[4] = {v[3].x, v[3].y, v[3].z, v[3].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = og * 4 + e;
      *reinterpret_cast<float4*>(lds + r * 128 + ((rg ^ swz(r)) << 4)) = make_float4(c0[e], c1[e], c2[e], c3[e]);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// one K tile of MFMA work for a wave: acc[2][2] += A(64 x BK) * B(64 x BK)^T
// ---------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void tile(const char* la, const char* lb, int wm, int wn, int lane, f32x16_t (&acc)[2][2]) {
    const int r = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + r;
        a[i] = *reinterpret_cast<const bf16x8_t*>(la + ra * 128 + (((ks * 2 + kg) ^ swz(ra)) << 4));
        const int rb = wn * 64 + i * 32 + r;
        b[i] = *reinterpret_cast<const bf16x8_t*>(lb + rb * 128 + (((ks * 2 + kg) ^ swz(rb)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void tile(const char* la, const char* lb, int wm, int wn, int lane, f32x16_t (&acc)[2][2]) {
    const int r = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = ks * 2 + kg;  // 0..31 within the tile
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + r;
        a[i] = *reinterpret_cast<const float*>(la + ra * 128 + (((k >> 2) ^ swz(ra)) << 4) + ((k & 3) << 2));
        const int rb = wn * 64 + i * 32 + r;
        b[i] = *reinterpret_cast<const float*>(lb + rb * 128 + (((k >> 2) ^ swz(rb)) << 4) + ((k & 3) << 2));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  int M, N, K, lda, ldb, ldc;
  int tiles_m, tiles_n, k_per_split;
  HeroGemmEpilogue epi;
};

template <typename T, int ALAY, int BLAY>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 64 KiB: 2 x (A 16K | B 16K); reused as the C tile
  constexpr int BK = Tr<T>::BK;

  // ---- workgroup -> (split, tile): XCD-contiguous chunks, then grouped along M for L2 reuse
  const int nwg = gridDim.x;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int ntile = g.tiles_m * g.tiles_n;
  const int split = wg / ntile;
  const int tile = wg - split * ntile;
  constexpr int GROUP = 8;
  const int per_group = GROUP * g.tiles_n;
  const int group = tile / per_group;
  const int first_m = group * GROUP;
  const int gsz = min(g.tiles_m - first_m, GROUP);
  const int in_group = tile - group * per_group;
  const int pid_m = first_m + in_group % gsz;
  const int pid_n = in_group / gsz;
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  Stage<T, ALAY> sa;
  Stage<T, BLAY> sb;
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    sa.load(A, g.lda, m0, g.M, kbeg, kend);
    sb.load(B, g.ldb, n0, g.N, kbeg, kend);
    sa.store(smem);
    sb.store(smem + 16384);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 32768;
    char* nxt = smem + ((kt + 1) & 1) * 32768;
    const bool more = kt + 1 < nk;
    if (more) {
      sa.load(A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend);
      sb.load(B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend);
    }
    Mma<T>::tile(cur, cur + 16384, wm, wn, lane, acc);
    if (more) {
      sa.store(nxt);
      sb.store(nxt + 16384);
    }
    __syncthreads();
  }

  const HeroGemmEpilogue& e = g.epi;
  // ---- split-K: fp32 atomics straight from the accumulator layout
  if (e.split_k > 1) {
    float* C = static_cast<float*>(g.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int gn = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < g.M && gn < g.N) atomicAdd(C + (size_t)gm * g.ldc + gn, acc[i][j][r]);
        }
      }
    return;
  }

  // ---- accumulators -> LDS C tile [128][128] fp32
  float* lc = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        lc[row * BN + col] = acc[i][j][r];
      }
    }
  __syncthreads();

  DropCtx drop(e.dropout);
  const T* R = static_cast<const T*>(e.residual);
  T* X = static_cast<T*>(e.aux);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int q = threadIdx.x + NT * it;
    const int row = q >> 5, c4 = (q & 31) * 4;
    const int gm = m0 + row, gn = n0 + c4;
    if (gm >= g.M || gn >= g.N) continue;
    float4 v = *reinterpret_cast<const float4*>(lc + row * BN + c4);
    const size_t off = (size_t)gm * g.ldc + gn;
    if (e.bias) {
      const float4 b = *reinterpret_cast<const float4*>(e.bias + gn);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (e.act == HERO_ACT_GELU) {
      V4<T>::st(X + off, v);
      v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
    } else if (e.act == HERO_ACT_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      if (X) V4<T>::st(X + off, v);
    } else if (e.act == HERO_ACT_GELU_BWD) {
      const float4 u = V4<T>::ld(X + off);
      v.x *= gelu_erf_grad(u.x); v.y *= gelu_erf_grad(u.y); v.z *= gelu_erf_grad(u.z); v.w *= gelu_erf_grad(u.w);
    } else if (e.act == HERO_ACT_RELU_BWD) {
      const float4 u = V4<T>::ld(X + off);
      v.x = u.x > 0.f ? v.x : 0.f; v.y = u.y > 0.f ? v.y : 0.f; v.z = u.z > 0.f ? v.z : 0.f; v.w = u.w > 0.f ? v.w : 0.f;
    }
    if (drop.on()) {
      const float4 mk = drop.mask4(((uint64_t)gm * (uint64_t)g.N + (uint64_t)gn) >> 2);
      v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
    }
    if (R) {
      const float4 rr = V4<T>::ld(R + off);
      v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
    }
    if (e.out_f32) {
      float* C = static_cast<float*>(g.C) + off;
      if (e.beta != 0.f) {
        const float4 o = *reinterpret_cast<const float4*>(C);
        v.x += e.beta * o.x; v.y += e.beta * o.y; v.z += e.beta * o.z; v.w += e.beta * o.w;
      }
      *reinterpret_cast<float4*>(C) = v;
    } else {
      V4<T>::st(static_cast<T*>(g.C) + off, v);
    }
  }
}

// out <- beta * out over an [M, N] fp32 matrix (pre-pass of the split-K atomics path)
__global__ void scale_f32_kernel(float* c, int M, int N, int ldc, float beta) {
  const size_t n4 = (size_t)N >> 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)M * n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / n4, c4 = (i - r * n4) * 4;
    float4* p = reinterpret_cast<float4*>(c + r * ldc + c4);
    if (beta == 0.f) {
      *p = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      float4 v = *p;
      v.x *= beta; v.y *= beta; v.z *= beta; v.w *= beta;
      *p = v;
    }
  }
}

// ---- optional per-launch timing with HIP events (bench.py roofline leg) ----------------------------
struct ProfSlot {
  std::vector<hipEvent_t> ev;  // start/stop pairs
  std::vector<double> flops;
};
static ProfSlot g_prof[8];
static bool g_prof_on = false;

template <typename T, int AL, int BL>
static int launch(const GemmArgs& g, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, AL, BL>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_set = true;
  }
  const int split = g.epi.split_k > 1 ? g.epi.split_k : 1;
  const int grid = g.tiles_m * g.tiles_n * split;
  ProfSlot* ps = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof_on) {
    ps = &g_prof[(sizeof(T) == 2 ? 4 : 0) + AL * 2 + BL];
    if (ps->flops.size() < 16384 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
      (void)hipEventRecord(e0, s);
    } else {
      ps = nullptr;
    }
  }
  hipLaunchKernelGGL((gemm_kernel<T, AL, BL>), dim3(grid), dim3(NT), 65536, s, g);
  if (ps) {
    (void)hipEventRecord(e1, s);
    ps->ev.push_back(e0);
    ps->ev.push_back(e1);
    ps->flops.push_back(2.0 * (double)g.M * (double)g.N * (double)g.K);
  }
  return check_launch("hero_gemm");
}

template <typename T>
static int dispatch(const GemmArgs& g, int al, int bl, hipStream_t s) {
  if (al == HERO_LAYOUT_K && bl == HERO_LAYOUT_K) return launch<T, HERO_LAYOUT_K, HERO_LAYOUT_K>(g, s);
  if (al == HERO_LAYOUT_K && bl == HERO_LAYOUT_O) return launch<T, HERO_LAYOUT_K, HERO_LAYOUT_O>(g, s);
  if (al == HERO_LAYOUT_O && bl == HERO_LAYOUT_O) return launch<T, HERO_LAYOUT_O, HERO_LAYOUT_O>(g, s);
  if (al == HERO_LAYOUT_O && bl == HERO_LAYOUT_K) return launch<T, HERO_LAYOUT_O, HERO_LAYOUT_K>(g, s);
  set_error("hero_gemm: bad layout (%d, %d)", al, bl);
  return HERO_ERR_ARG;
}

}  // namespace hero

using namespace hero;

extern "C" int hero_gemm(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                         int a_layout, int b_layout, int dtype, const HeroGemmEpilogue* epi, hero_stream_t stream) {
  HERO_REQUIRE(A && B && C && epi, "hero_gemm: null pointer");
  HERO_REQUIRE(dtype == HERO_F32 || dtype == HERO_BF16, "hero_gemm: bad dtype %d", dtype);
  if (M <= 0 || N <= 0) return HERO_OK;
  HERO_REQUIRE(K >= 0, "hero_gemm: K < 0");
  const int vec = dtype == HERO_BF16 ? 8 : 4;
  HERO_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "hero_gemm: N (%d) and ldc (%d) must be multiples of 4", N, ldc);
  HERO_REQUIRE(lda % vec == 0 && ldb % vec == 0, "hero_gemm: lda/ldb (%d, %d) must be multiples of %d", lda, ldb, vec);
  HERO_REQUIRE(a_layout == HERO_LAYOUT_K ? K % vec == 0 : M % vec == 0,
               "hero_gemm: A contiguous dim must be a multiple of %d (M=%d K=%d layout=%d)", vec, M, K, a_layout);
  HERO_REQUIRE(b_layout == HERO_LAYOUT_K ? K % vec == 0 : N % vec == 0,
               "hero_gemm: B contiguous dim must be a multiple of %d (N=%d K=%d layout=%d)", vec, N, K, b_layout);
  HERO_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "hero_gemm: operands must be 16-byte aligned");
  HERO_REQUIRE(!epi->residual || (((uintptr_t)epi->residual) & 7) == 0, "hero_gemm: residual misaligned");
  HERO_REQUIRE(!(epi->act == HERO_ACT_GELU || epi->act == HERO_ACT_GELU_BWD || epi->act == HERO_ACT_RELU_BWD) || epi->aux,
               "hero_gemm: activation %d needs aux", epi->act);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  g.epi = *epi;
  const int bk = dtype == HERO_BF16 ? 64 : 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (epi->split_k > 1) {
    HERO_REQUIRE(epi->out_f32 && epi->act == HERO_ACT_NONE && !epi->bias && !epi->residual && epi->dropout.threshold16 == 0,
                 "hero_gemm: split_k supports only a plain fp32 accumulate epilogue");
    const int ktiles = (K + bk - 1) / bk;
    int split = epi->split_k < ktiles ? epi->split_k : ktiles;
    if (split > 1) {
      const int per = (ktiles + split - 1) / split;
      split = (ktiles + per - 1) / per;
      if (split > 1) {
        if (epi->beta != 1.f) {  // atomics accumulate into beta * C
          hipLaunchKernelGGL(scale_f32_kernel, dim3(1024), dim3(256), 0, s, static_cast<float*>(C), M, N, ldc, epi->beta);
          const int rc = check_launch("hero_gemm(scale)");
          if (rc) return rc;
        }
        g.k_per_split = per * bk;
        g.epi.split_k = split;
        return dtype == HERO_BF16 ? dispatch<bf16_t>(g, a_layout, b_layout, s) : dispatch<float>(g, a_layout, b_layout, s);
      }
    }
  }
  g.k_per_split = K > 0 ? ((K + bk - 1) / bk) * bk : bk;
  g.epi.split_k = 1;
  return dtype == HERO_BF16 ? dispatch<bf16_t>(g, a_layout, b_layout, s) : dispatch<float>(g, a_layout, b_layout, s);
}

// Per-launch timing of the GEMM kernels with HIP events on the launch stream (used by bench.py
// for the roofline leg). slot = (dtype == BF16 ? 4 : 0) + a_layout * 2 + b_layout.
extern "C" int hero_prof_enable(int on) {
  for (auto& p : g_prof) {
    for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
    p.ev.clear();
    p.flops.clear();
  }
  g_prof_on = on != 0;
  return HERO_OK;
}
extern "C" int hero_prof_read(int slot, double* total_ms, double* total_flops, long long* launches) {
  HERO_REQUIRE(slot >= 0 && slot < 8 && total_ms && total_flops && launches, "hero_prof_read: bad arguments");
  ProfSlot& p = g_prof[slot];
  double ms = 0.0, fl = 0.0;
  for (size_t i = 0; i < p.flops.size(); ++i) {
    float t = 0.f;
    if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) {
      set_error("hero_prof_read: event query failed");
      return HERO_ERR_LAUNCH;
    }
    ms += t;
    fl += p.flops[i];
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = (long long)p.flops.size();
  return HERO_OK;
}
