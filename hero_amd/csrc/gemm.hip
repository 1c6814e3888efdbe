// MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   C[M,N] = op(A) * op(B)     bf16: v_mfma_f32_32x32x16_bf16, BK = 64
//                              f32 : v_mfma_f32_32x32x2_f32 (exact f32), BK = 32
//
// Geometry is a template: WM x WN waves per workgroup, each wave owning TM x TN MFMA 32x32 tiles.
//   Cfg<2,2,2,2>  128 x 128, 4 waves,  64 KiB LDS (2 workgroups / CU)
//   Cfg<2,4,4,2>  256 x 256, 8 waves, 128 KiB LDS
//
// LDS image of an operand tile is always [rows][128 bytes] with the eight 16-byte chunks of a row
// XOR-swizzled by swz(row) = (row ^ row>>3) & 7 (measured: SQ_LDS_BANK_CONFLICT = 0), whatever the
// operand's layout in HBM:
//   HERO_LAYOUT_K (reduction contiguous)  16 B global -> 16 B LDS
//   HERO_LAYOUT_O (outer dim contiguous)  4 reduction rows x 16 B are transposed in registers and
//                                         written as 8 B (bf16) / 16 B (f32) pieces
// so forward (x W^T), dgrad (dY W) and wgrad (dY^T X) share one main loop and need no transposed
// copies of weights or activations in HBM.
// Staging is register-staged and split: loads for tile t+1 are issued before the MFMAs of tile t and
// written to LDS after them; double-buffered LDS, one barrier per K tile.  All per-lane addressing
// is computed once: a 32-bit element offset from a wave-uniform base pointer that the k-loop
// advances with scalar arithmetic.  Rows outside the matrix are clamped to a valid row (they only
// feed outputs that are never stored), so full tiles need no masking; only a partial last K tile
// takes the masked path.  (A branch around a load makes hipcc wait vmcnt(0) per load.)
// The epilogue goes through LDS (128 output rows at a time): bias is loaded once per thread, the
// residual / aux / C loads of four rows are issued together, stores are 8-16 B per lane along N.
#include "gemm_args.h"
#include <vector>

namespace hero {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

template <typename T> struct Tr;
template <> struct Tr<bf16_t> { static constexpr int BK = 64, EPC = 8; };
template <> struct Tr<float>  { static constexpr int BK = 32, EPC = 4; };

template <int WM_, int WN_, int TM_, int TN_> struct Cfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int EPI_ROWS = BM < 128 ? BM : (BM % 128 == 0 ? 128 : BM / 2);   // output rows per epilogue pass
  static constexpr int EPI_BYTES = EPI_ROWS * BN * 4;                 // one pass of fp32
  static constexpr int LDS = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
};

// ---------------------------------------------------------------------------------------------
// global -> register -> LDS staging of one R x BK operand tile by NT threads
// ---------------------------------------------------------------------------------------------
template <typename T, int LAY, int R, int NT> struct Stage;

template <typename T, int R, int NT> struct Stage<T, HERO_LAYOUT_K, R, NT> {
  static constexpr int N = R * 8 / NT;
  uint4 v[N];
  int goff[N];    // element offset from the tile's uniform base (row0, k0)
  int loff[N];    // LDS byte offset
  int kcol[N];    // k offset (elements) of this chunk inside the tile, for the tail mask
  __device__ __forceinline__ void init(int ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int q = threadIdx.x + NT * i;
      const int r = q >> 3, c = q & 7;
      const int cr = min(row0 + r, nrows - 1) - row0;
      goff[i] = cr * ld + c * Tr<T>::EPC;
      loff[i] = r * 128 + ((c ^ swz(r)) << 4);
      kcol[i] = c * Tr<T>::EPC;
    }
  }
  static __device__ __forceinline__ const T* base(const T* p, int ld, int row0, int k) { return p + (size_t)row0 * ld + k; }
  static __device__ __forceinline__ const T* advance(const T* b, int) { return b + Tr<T>::BK; }
  __device__ __forceinline__ void load(const T* __restrict__ b) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const uint4*>(b + goff[i]);
  }
  __device__ __forceinline__ void load_tail(const T* __restrict__ b, int krem) {   // krem = valid k in this tile
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int back = kcol[i] < krem ? 0 : kcol[i] - (krem - Tr<T>::EPC);           // stay inside the row
      v[i] = *reinterpret_cast<const uint4*>(b + goff[i] - back);
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
    for (int i = 0; i < N; ++i) *reinterpret_cast<uint4*>(lds + loff[i]) = v[i];
  }
  __device__ __forceinline__ void store_tail(char* lds, int krem) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const bool ok = kcol[i] < krem;
      uint4 t = v[i];
      t.x = ok ? t.x : 0u; t.y = ok ? t.y : 0u; t.z = ok ? t.z : 0u; t.w = ok ? t.w : 0u;
      *reinterpret_cast<uint4*>(lds + loff[i]) = t;
    }
  }
};

// outer-contiguous operands: thread slot = (out group of EPC elements) x (reduction group of 4); 2R slots.
// The 4 x EPC block is transposed in registers and written as EPC pieces of 4 reduction elements.
template <typename T, int R, int NT> struct Stage<T, HERO_LAYOUT_O, R, NT> {
  static constexpr int EPC = Tr<T>::EPC;
  static constexpr int N = (2 * R + NT - 1) / NT, OG = R / EPC;   // threads beyond 2R slots idle
  uint4 v[N][4];
  int goff[N];      // element offset of (reduction row rg*4, outer og*EPC) from the uniform base
  int loff[N];      // LDS byte offset of outer row og*EPC, reduction group rg (before the per-row swizzle)
  int krow[N];      // first reduction row of the slot inside the tile
  int ld_;
  __device__ __forceinline__ void init(int ld, int row0, int nrows) {
    ld_ = ld;
#pragma unroll
    for (int s = 0; s < N; ++s) {
      const int q = threadIdx.x + NT * s;
      const int og = q % OG, rg = q / OG;
      const int co = min(row0 + og * EPC, nrows - EPC) - row0;
      goff[s] = rg * 4 * ld + co;
      loff[s] = og * EPC * 128 + rg * (sizeof(T) * 4);      // + swizzle of the chunk index per row in store()
      krow[s] = (2 * R < NT * N && q >= 2 * R) ? 1 << 20 : rg * 4;   // idle slot: never valid
    }
  }
  static __device__ __forceinline__ const T* base(const T* p, int ld, int row0, int k) { return p + (size_t)k * ld + row0; }
  static __device__ __forceinline__ const T* advance(const T* b, int ld) { return b + (size_t)Tr<T>::BK * ld; }
  __device__ __forceinline__ void load(const T* __restrict__ b) {
#pragma unroll
    for (int s = 0; s < N; ++s) {
      if (2 * R < NT * N && krow[s] >= (1 << 20)) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[s][j] = *reinterpret_cast<const uint4*>(b + goff[s] + j * ld_);
    }
  }
  __device__ __forceinline__ void load_tail(const T* __restrict__ b, int krem) {
#pragma unroll
    for (int s = 0; s < N; ++s) {
      if (2 * R < NT * N && krow[s] >= (1 << 20)) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kr = krow[s] + j;
        const int back = kr < krem ? 0 : kr - (krem - 1);                           // clamp to the last valid row
        v[s][j] = *reinterpret_cast<const uint4*>(b + goff[s] + (j - back) * ld_);
      }
    }
  }
  __device__ __forceinline__ void store_impl(char* lds, int krem) const {
#pragma unroll
    for (int s = 0; s < N; ++s) {
      if (2 * R < NT * N && krow[s] >= (1 << 20)) continue;
      uint32_t w[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t m = (krow[s] + j < krem) ? 0xffffffffu : 0u;
        w[j][0] = v[s][j].x & m; w[j][1] = v[s][j].y & m; w[j][2] = v[s][j].z & m; w[j][3] = v[s][j].w & m;
      }
      const int r0 = loff[s] >> 7;                 // og * EPC
      const int inrow = loff[s] & 127;             // byte position of the reduction group in a row
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int r = r0 + e;
        char* dst = lds + r * 128 + ((((inrow >> 4) ^ swz(r)) << 4) | (inrow & 15));
        if (sizeof(T) == 2) {
          const int c = e >> 1;
          uint2 o;
          if (e & 1) {
            o.x = (w[0][c] >> 16) | (w[1][c] & 0xffff0000u);
            o.y = (w[2][c] >> 16) | (w[3][c] & 0xffff0000u);
          } else {
            o.x = (w[0][c] & 0xffffu) | (w[1][c] << 16);
            o.y = (w[2][c] & 0xffffu) | (w[3][c] << 16);
          }
          *reinterpret_cast<uint2*>(dst) = o;
        } else {
          *reinterpret_cast<uint4*>(dst) = make_uint4(w[0][e], w[1][e], w[2][e], w[3][e]);
        }
      }
    }
  }
  __device__ __forceinline__ void store(char* lds) const { store_impl(lds, 1 << 19); }
  __device__ __forceinline__ void store_tail(char* lds, int krem) const { store_impl(lds, krem); }
};

// ---------------------------------------------------------------------------------------------
// one K tile of MFMA work for a wave: acc[TM][TN] += A(TM*32 x BK) * B(TN*32 x BK)^T
// ---------------------------------------------------------------------------------------------
template <typename T, int TM, int TN> struct Mma;
template <int TM, int TN> struct Mma<bf16_t, TM, TN> {
  static __device__ __forceinline__ void tile(const char* la, const char* lb, int arow0, int brow0, int lane, f32x16_t (&acc)[TM][TN]) {
    const int r = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + kg;
      bf16x8_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ra = arow0 + i * 32 + r;
        a[i] = *reinterpret_cast<const bf16x8_t*>(la + ra * 128 + ((kc ^ swz(ra)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int rb = brow0 + j * 32 + r;
        b[j] = *reinterpret_cast<const bf16x8_t*>(lb + rb * 128 + ((kc ^ swz(rb)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};
template <int TM, int TN> struct Mma<float, TM, TN> {
  static __device__ __forceinline__ void tile(const char* la, const char* lb, int arow0, int brow0, int lane, f32x16_t (&acc)[TM][TN]) {
    const int r = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = ks * 2 + kg;  // 0..31 within the tile
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ra = arow0 + i * 32 + r;
        a[i] = *reinterpret_cast<const float*>(la + ra * 128 + (((k >> 2) ^ swz(ra)) << 4) + ((k & 3) << 2));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int rb = brow0 + j * 32 + r;
        b[j] = *reinterpret_cast<const float*>(lb + rb * 128 + (((k >> 2) ^ swz(rb)) << 4) + ((k & 3) << 2));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
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
  int group;            // M-tiles per L2 locality group (tile_coord)
  HeroGemmEpilogue epi;
};

// ---------------------------------------------------------------------------------------------
// epilogue shared by both staging flavours
// ---------------------------------------------------------------------------------------------
// EK selects the epilogue features at COMPILE time for the hot-path combinations (the generic
// runtime-flag version costs thousands of instructions per thread); EK_GENERIC keeps every flag.

template <typename T, typename CF, int EK>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16_t (&acc)[CF::TM][CF::TN], char* smem, int m0, int n0,
                                              int arow0, int brow0, int lane, int kbeg = 0) {
  constexpr int BM = CF::BM, BN = CF::BN, NT = CF::NT, TM = CF::TM, TN = CF::TN;
  constexpr bool GEN = (EK & EK_GENERIC) != 0;
  const HeroGemmEpilogue& e = g.epi;
  // ---- split-K: fp32 straight from the accumulator layout - atomics into C, or (split_stride != 0) plain stores into
  // this split's own slab C + split * split_stride, folded afterwards in slab order (hero_fold_slabs): bit-reproducible
  if (GEN && e.split_k > 1) {
    const bool slab = e.split_stride != 0;
    float* C = static_cast<float*>(g.C) + (slab ? (size_t)(kbeg / g.k_per_split) * (size_t)e.split_stride : (size_t)0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int gn = n0 + brow0 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = m0 + arow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < g.M && gn < g.N) {
            if (slab) C[(size_t)gm * g.ldc + gn] = acc[i][j][r];
            else atomicAdd(C + (size_t)gm * g.ldc + gn, acc[i][j][r]);
          }
        }
      }
    return;
  }

  // ---- accumulators -> LDS (128 rows x BN fp32 per pass) -> vectorised epilogue
  float* lc = reinterpret_cast<float*>(smem);
  DropCtx drop(e.dropout);
  const T* R = (GEN || (EK & EK_RES)) ? static_cast<const T*>(e.residual) : nullptr;
  T* X = static_cast<T*>(e.aux);
  // the specialised instantiations serve two activation codes each (a uniform run-time choice inside the same epilogue):
  // GELU / GELU_DG (what is saved: the pre-activation or the derivative), GELU_BWD / MUL_AUX (what is read back)
  const int act = GEN ? e.act
                      : ((EK & EK_GELU) ? (e.act == HERO_ACT_GELU_DG ? HERO_ACT_GELU_DG : HERO_ACT_GELU)
                                        : ((EK & EK_GELU_BWD) ? (e.act == HERO_ACT_MUL_AUX ? HERO_ACT_MUL_AUX : HERO_ACT_GELU_BWD) : HERO_ACT_NONE));
  const bool use_bias = GEN ? (e.bias != nullptr) : ((EK & EK_BIAS) != 0);
  const bool use_drop = (GEN || (EK & EK_DROP)) && drop.on();
  const bool out_f32 = GEN && e.out_f32;
  constexpr int PR = CF::EPI_ROWS;                  // rows per pass
  constexpr int PASSES = BM / PR;
  constexpr int C4 = BN / 4;                        // float4 per row
  constexpr int ITERS = PR * C4 / NT;
  // this thread's column group is the same in every iteration (NT is a multiple of C4)
  const int c4 = (threadIdx.x % C4) * 4;
  const int gn = n0 + c4;
  const bool col_ok = gn < g.N;
  const int gnc = min(gn, g.N - 4);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (use_bias) bias4 = *reinterpret_cast<const float4*>(e.bias + gnc);
  const bool ld_aux = act == HERO_ACT_GELU_BWD || act == HERO_ACT_RELU_BWD || act == HERO_ACT_MUL_AUX;
  const bool ld_c = out_f32 && e.beta != 0.f;
  constexpr int RSTEP = NT / C4;                 // rows covered per iteration
  // The specialised bf16 epilogues fetch the residual / saved pre-activation of the WHOLE pass up
  // front (raw 8-byte loads, issued before the accumulators are staged, so their HBM latency hides
  // behind the staging and its barrier); batches of 4 left ~4 exposed round trips per tile.
  constexpr bool PRE = !GEN && sizeof(T) == 2;
  constexpr int UN = PRE ? ITERS : 4;
  // column sums of the stored values (bias gradient of the producing layer): generic + gelu' variants
  constexpr bool CSUM = GEN || (EK & EK_GELU_BWD);
  const bool do_csum = CSUM && e.colsum != nullptr;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int pass = 0; pass < PASSES; ++pass) {
    uint2 pre_r[PRE ? ITERS : 1], pre_u[PRE ? ITERS : 1];
    if constexpr (PRE) {
#pragma unroll
      for (int u = 0; u < ITERS; ++u) {
        const int gm = m0 + pass * PR + threadIdx.x / C4 + u * RSTEP;
        const size_t off = (size_t)min(gm, g.M - 1) * g.ldc + gnc;
        if (R) pre_r[u] = *reinterpret_cast<const uint2*>(R + off);
        if (ld_aux) pre_u[u] = *reinterpret_cast<const uint2*>(X + off);
      }
    }
    if (pass) __syncthreads();
    if (arow0 / PR == pass) {
      const int rbase = arow0 - pass * PR;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = brow0 + j * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            lc[row * BN + col] = acc[i][j][r];
          }
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int it0 = 0; it0 < ITERS; it0 += UN) {
      size_t off[UN];
      bool ok[UN];
      float4 rr[UN], uu[UN], cc[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {               // all global loads of the group first (batched)
        const int row = threadIdx.x / C4 + (it0 + u) * RSTEP;
        const int gm = m0 + pass * PR + row;
        ok[u] = col_ok && gm < g.M;
        off[u] = (size_t)min(gm, g.M - 1) * g.ldc + gnc;
        if constexpr (PRE) {
          if (R) rr[u] = make_float4(__uint_as_float(pre_r[u].x << 16), __uint_as_float(pre_r[u].x & 0xffff0000u),
                                     __uint_as_float(pre_r[u].y << 16), __uint_as_float(pre_r[u].y & 0xffff0000u));
          if (ld_aux) uu[u] = make_float4(__uint_as_float(pre_u[u].x << 16), __uint_as_float(pre_u[u].x & 0xffff0000u),
                                          __uint_as_float(pre_u[u].y << 16), __uint_as_float(pre_u[u].y & 0xffff0000u));
        } else {
          if (R) rr[u] = V4<T>::ld(R + off[u]);
          if (ld_aux) uu[u] = V4<T>::ld(X + off[u]);
        }
        if (ld_c) cc[u] = *reinterpret_cast<const float4*>(static_cast<const float*>(g.C) + off[u]);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int row = threadIdx.x / C4 + (it0 + u) * RSTEP;
        float4 v = *reinterpret_cast<const float4*>(lc + row * BN + c4);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (act == HERO_ACT_GELU) {
          if (ok[u]) V4<T>::st(X + off[u], v);
          v.x = gelu_fwd<T>(v.x); v.y = gelu_fwd<T>(v.y); v.z = gelu_fwd<T>(v.z); v.w = gelu_fwd<T>(v.w);
        } else if (act == HERO_ACT_GELU_DG) {
          float4 dg;
          gelu_both<T>(v.x, v.x, dg.x); gelu_both<T>(v.y, v.y, dg.y); gelu_both<T>(v.z, v.z, dg.z); gelu_both<T>(v.w, v.w, dg.w);
          if (ok[u]) V4<T>::st(X + off[u], dg);
        } else if (act == HERO_ACT_MUL_AUX) {
          v.x *= uu[u].x; v.y *= uu[u].y; v.z *= uu[u].z; v.w *= uu[u].w;
        } else if (GEN && act == HERO_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          if (X && ok[u]) V4<T>::st(X + off[u], v);
        } else if (act == HERO_ACT_GELU_BWD) {
          v.x *= gelu_grad<T>(uu[u].x); v.y *= gelu_grad<T>(uu[u].y); v.z *= gelu_grad<T>(uu[u].z); v.w *= gelu_grad<T>(uu[u].w);
        } else if (GEN && act == HERO_ACT_RELU_BWD) {
          v.x = uu[u].x > 0.f ? v.x : 0.f; v.y = uu[u].y > 0.f ? v.y : 0.f; v.z = uu[u].z > 0.f ? v.z : 0.f; v.w = uu[u].w > 0.f ? v.w : 0.f;
        }
        if (use_drop) {
          const int gm = m0 + pass * PR + row;
          const float4 mk = drop.mask4(((uint64_t)gm * (uint64_t)g.N + (uint64_t)gn) >> 2);
          v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
        }
        if (R) { v.x += rr[u].x; v.y += rr[u].y; v.z += rr[u].z; v.w += rr[u].w; }
        if (do_csum && ok[u]) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }
        if (out_f32) {
          if (ld_c) { v.x += e.beta * cc[u].x; v.y += e.beta * cc[u].y; v.z += e.beta * cc[u].z; v.w += e.beta * cc[u].w; }
          if (ok[u]) *reinterpret_cast<float4*>(static_cast<float*>(g.C) + off[u]) = v;
        } else {
          if (ok[u]) V4<T>::st(static_cast<T*>(g.C) + off[u], v);
        }
      }
    }
  }
  if (do_csum) {                                   // fold the NT / C4 row groups through LDS, one atomic per column
    __syncthreads();
    *reinterpret_cast<float4*>(lc + (threadIdx.x / C4) * BN + c4) = cs;
    __syncthreads();
    if (threadIdx.x < C4 && col_ok) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < RSTEP; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(lc + k * BN + c4);
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
      }
      if (e.colsum_partial) {                      // deterministic partial table [ceil(M / 64), N] (hero_hip.h): row m0 / 64, zeros below
        float* pr = e.colsum + (size_t)(m0 >> 6) * g.N + gn;
        const int nb = min(BM >> 6, ((g.M + 63) >> 6) - (m0 >> 6));
        *reinterpret_cast<float4*>(pr) = t;
        for (int b = 1; b < nb; ++b) *reinterpret_cast<float4*>(pr + (size_t)b * g.N) = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        atomicAdd(e.colsum + gn, t.x); atomicAdd(e.colsum + gn + 1, t.y); atomicAdd(e.colsum + gn + 2, t.z); atomicAdd(e.colsum + gn + 3, t.w);
      }
    }
  }
}

// workgroup -> (split, tile): XCD-contiguous chunks, then grouped along M for L2 reuse
struct TileCoord { int m0, n0, kbeg, kend; };
template <typename CF>
__device__ __forceinline__ TileCoord tile_coord(const GemmArgs& g) {
  const int nwg = gridDim.x;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int ntile = g.tiles_m * g.tiles_n;
  const int split = wg / ntile;
  const int tile = wg - split * ntile;
  const int GROUP = g.group;
  const int per_group = GROUP * g.tiles_n;
  const int group = tile / per_group;
  const int first_m = group * GROUP;
  const int gsz = min(g.tiles_m - first_m, GROUP);
  const int in_group = tile - group * per_group;
  TileCoord t;
  t.m0 = (first_m + in_group % gsz) * CF::BM;
  t.n0 = (in_group / gsz) * CF::BN;
  t.kbeg = split * g.k_per_split;
  t.kend = min(g.K, t.kbeg + g.k_per_split);
  return t;
}

// ---------------------------------------------------------------------------------------------
// direct-to-LDS staging (global_load_lds, 16 B per lane) for K-contiguous operands, K % BK == 0.
// One wave instruction fills 8 consecutive tile rows (8 x 128 B, lane-linear in LDS); the chunk
// swizzle is applied on the SOURCE address (lane (row, c') fetches chunk c' ^ swz(row)).
// No staging VGPRs, no ds_write: the VGPR->LDS write path is what bounds the register-staged loop.
// ---------------------------------------------------------------------------------------------
template <typename T, int R, int NT>
struct GldsStage {
  static constexpr int PER_WAVE = R / 8 / (NT / 64);     // instructions per wave per tile
  int goff[PER_WAVE];                                      // element offset from the uniform base
  int lrow0;                                               // first tile row of this wave's first instruction
  __device__ __forceinline__ void init(int ld, int row0, int nrows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lrow0 = wave * (R / (NT / 64));
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int r = lrow0 + i * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz(r);
      goff[i] = (min(row0 + r, nrows - 1) - row0) * ld + c * Tr<T>::EPC;
    }
  }
  __device__ __forceinline__ void issue(const T* __restrict__ b, char* lds) const {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i)
      __builtin_amdgcn_global_load_lds(b + goff[i], (__attribute__((address_space(3))) void*)(lds + (lrow0 + i * 8) * 128), 16, 0, 0);
  }
};

template <typename CF> struct GldsRing {
  static constexpr int NS = CF::BM * CF::BN <= 64 * 64 ? 4 : 2;
  static constexpr int LDS = NS * CF::STAGE > CF::EPI_BYTES ? NS * CF::STAGE : CF::EPI_BYTES;
};

template <typename T, typename CF, int EK>
__global__ __launch_bounds__(CF::NT) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_glds_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = Tr<T>::BK, BM = CF::BM, BN = CF::BN, NT = CF::NT, TM = CF::TM, TN = CF::TN;
  const TileCoord tc = tile_coord<CF>(g);
  const int m0 = tc.m0, n0 = tc.n0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / CF::WN, wn = wave % CF::WN;
  const int arow0 = wm * TM * 32, brow0 = wn * TN * 32;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  GldsStage<T, BM, NT> sa;
  GldsStage<T, BN, NT> sb;
  sa.init(g.lda, m0, g.M);
  sb.init(g.ldb, n0, g.N);
  const T* pa = static_cast<const T*>(g.A) + (size_t)m0 * g.lda + tc.kbeg;
  const T* pb = static_cast<const T*>(g.B) + (size_t)n0 * g.ldb + tc.kbeg;
  const int nk = (tc.kend - tc.kbeg) / BK;
  // Ring of NSG stages, NSG - 1 of them in flight (counted vmcnt).  The 64 x 64 tiles serve the problems that are too
  // small to fill the chip (the 1920-row Temporal Transformer / query shapes): one or two workgroups per CU and only
  // 8 MFMAs per wave and step, so a step costs the load round trip divided by the stages in flight - 0.67 us with
  // one stage ahead (48 steps at K = 3072 = 32 us).  Larger tiles keep two stages (64-80 KiB, two workgroups / CU).
  constexpr int NSG = GldsRing<CF>::NS;
  constexpr int PW = GldsStage<T, BM, NT>::PER_WAVE + GldsStage<T, BN, NT>::PER_WAVE;
  static_assert((NSG - 1) * PW <= 60, "vmcnt range");
  auto wait_stages = [](int later) {            // at most `later` whole stages may still be in flight
    if (NSG > 3 && later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
    else if (NSG > 2 && later >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const T* qa = pa;
  const T* qb = pb;
#pragma unroll
  for (int d = 0; d < NSG - 1; ++d)
    if (d < nk) {
      sa.issue(qa, smem + d * CF::STAGE);
      sb.issue(qb, smem + d * CF::STAGE + CF::A_BYTES);
      qa += BK;
      qb += BK;
    }
  // An LDS-DMA is ordered for other waves' ds_reads only by the issuing wave's vmcnt + a barrier.
  wait_stages(min(NSG - 2, nk - 1));
  __syncthreads();                                   // tile 0 has landed
  int cs = 0, fs = NSG - 1;                          // slot being read / slot to fill (the one read in the previous step)
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + cs * CF::STAGE;
    if (kt + NSG - 1 < nk) {
      sa.issue(qa, smem + fs * CF::STAGE);
      sb.issue(qb, smem + fs * CF::STAGE + CF::A_BYTES);
      qa += BK;
      qb += BK;
    }
    cs = cs + 1 == NSG ? 0 : cs + 1;
    fs = fs + 1 == NSG ? 0 : fs + 1;
    Mma<T, TM, TN>::tile(cur, cur + CF::A_BYTES, arow0, brow0, lane, acc);
    wait_stages(min(NSG - 2, nk - 2 - kt));          // tile kt + 1 landed
    __syncthreads();                                 // ... and everyone is done with cur
  }
  gemm_epilogue<T, CF, EK>(g, acc, smem, m0, n0, arow0, brow0, lane, tc.kbeg);
}

// ---------------------------------------------------------------------------------------------
// wgrad flavour (both operands OUTER-contiguous: dY[M, N_out], X[M, K_in], reduction over rows):
// direct-to-LDS staging of the untransposed [64 k][128 outer] tiles + ds_read_b64_tr_b16 fragment
// reads.  Measured lane map of the transpose read (tools/lab/trprobe.hip): in each 16-lane group
// lane i receives column i (4 consecutive rows) of the 4 x 16 block whose 8-byte row segments are
// addressed by the group's lanes (lane p -> row p>>2, columns 4*(p&3)..+3).  Two reads give the 8
// consecutive k of one outer index = one 32x32x16 MFMA operand.  Swizzle: 16-B chunk index XOR
// 4*(k&3) (applied on the source address of the DMA and on the read address) spreads the 4 rows of
// a group over distinct bank ranges.  bf16 only, 128x128 tile, reduction length % 64 == 0.
// ---------------------------------------------------------------------------------------------
struct GldsStageO {
  int goff[4];
  int lrow0;
  __device__ __forceinline__ void init(int ld, int col0, int ncols) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rsub = lane >> 4, cp = lane & 15;
    lrow0 = wave * 16;
    const int c = cp ^ (4 * rsub);                       // (k & 3) == rsub for every row this lane touches
    const int oc = min(col0 + c * 8, ncols - 8) - col0;
#pragma unroll
    for (int i = 0; i < 4; ++i) goff[i] = (lrow0 + i * 4 + rsub) * ld + oc;
  }
  __device__ __forceinline__ void issue(const bf16_t* __restrict__ b, char* lds) const {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(b + goff[i], (__attribute__((address_space(3))) void*)(lds + (lrow0 + i * 4) * 256), 16, 0, 0);
  }
  // partial last tile: rows >= krem are fetched from the last valid row (then zeroed in LDS)
  __device__ __forceinline__ void issue_tail(const bf16_t* __restrict__ b, char* lds, int krem, int ld) const {
    const int rsub = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow0 + i * 4 + rsub;
      const int back = row < krem ? 0 : row - (krem - 1);
      __builtin_amdgcn_global_load_lds(b + goff[i] - back * ld, (__attribute__((address_space(3))) void*)(lds + (lrow0 + i * 4) * 256), 16, 0, 0);
    }
  }
};

// zero rows [krem, 64) of both operand images of a stage (all DMAs have landed, barrier passed)
__device__ __forceinline__ void zero_tail_rows(char* stage, int krem) {
  const int n = (64 - krem) * 16;
  for (int idx = threadIdx.x; idx < n; idx += 256) {
    const int off = (krem + (idx >> 4)) * 256 + (idx & 15) * 16;
    *reinterpret_cast<uint4*>(stage + off) = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>(stage + 16384 + off) = make_uint4(0, 0, 0, 0);
  }
}

// the 8 transpose reads of one k-step (2 A fragments + 2 B fragments, two halves each) + wait
__device__ __forceinline__ void tr_frags(unsigned a0, unsigned a1, unsigned b0, unsigned b1, int ks, bf16x8_t (&a)[2], bf16x8_t (&b)[2]) {
  uint2 r0, r1, r2, r3, r4, r5, r6, r7;
  const unsigned o = ks * 4096;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
      "ds_read_b64_tr_b16 %2, %9\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:1024\n\t"
      "ds_read_b64_tr_b16 %4, %10\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:1024\n\t"
      "ds_read_b64_tr_b16 %6, %11\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:1024\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
      : "v"(a0 + o), "v"(a1 + o), "v"(b0 + o), "v"(b1 + o)
      : "memory");
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 t;
  t = u32x4{r0.x, r0.y, r1.x, r1.y}; a[0] = __builtin_bit_cast(bf16x8_t, t);
  t = u32x4{r2.x, r2.y, r3.x, r3.y}; a[1] = __builtin_bit_cast(bf16x8_t, t);
  t = u32x4{r4.x, r4.y, r5.x, r5.y}; b[0] = __builtin_bit_cast(bf16x8_t, t);
  t = u32x4{r6.x, r6.y, r7.x, r7.y}; b[1] = __builtin_bit_cast(bf16x8_t, t);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_glds_tr_kernel(GemmArgs g) {
  typedef Cfg<2, 2, 2, 2> CF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileCoord tc = tile_coord<CF>(g);
  const int m0 = tc.m0, n0 = tc.n0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int arow0 = wm * 64, brow0 = wn * 64;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  GldsStageO sa, sb;
  sa.init(g.lda, m0, g.M);
  sb.init(g.ldb, n0, g.N);
  const bf16_t* pa = static_cast<const bf16_t*>(g.A) + (size_t)tc.kbeg * g.lda + m0;
  const bf16_t* pb = static_cast<const bf16_t*>(g.B) + (size_t)tc.kbeg * g.ldb + n0;
  const int nk = (tc.kend - tc.kbeg + 63) / 64;
  const int krem = (tc.kend - tc.kbeg) - (nk - 1) * 64;     // valid reduction rows in the last tile
  const bool has_tail = krem != 64;

  // per-lane read offsets inside a tile image [64 k][256 B]
  const int p = lane & 15, gq = (lane >> 4) & 1, kg = lane >> 5;
  unsigned fo[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int col = (f < 2 ? arow0 : brow0) + (f & 1) * 32 + gq * 16 + 4 * (p & 3);
    fo[f] = (kg * 8 + (p >> 2)) * 256 + ((((col >> 3) ^ (4 * (p >> 2))) << 4) | ((col & 7) * 2));
  }
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  if (nk > 0) {
    if (nk == 1 && has_tail) {
      sa.issue_tail(pa, smem, krem, g.lda);
      sb.issue_tail(pb, smem + 16384, krem, g.ldb);
    } else {
      sa.issue(pa, smem);
      sb.issue(pb, smem + 16384);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nk == 1 && has_tail) {
    zero_tail_rows(smem, krem);
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned cur = lds0 + (kt & 1) * 32768;
    char* nxt = smem + ((kt + 1) & 1) * 32768;
    const bool tail_next = has_tail && kt + 2 == nk;
    if (kt + 1 < nk) {
      pa += (size_t)64 * g.lda;
      pb += (size_t)64 * g.ldb;
      if (tail_next) {
        sa.issue_tail(pa, nxt, krem, g.lda);
        sb.issue_tail(pb, nxt + 16384, krem, g.ldb);
      } else {
        sa.issue(pa, nxt);
        sb.issue(pb, nxt + 16384);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t a[2], b[2];
      tr_frags(cur + fo[0], cur + fo[1], cur + 16384 + fo[2], cur + 16384 + fo[3], ks, a, b);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tail_next) {                                  // uniform: only before the last, partial tile
      zero_tail_rows(nxt, krem);
      __syncthreads();
    }
  }
  gemm_epilogue<bf16_t, CF, EK_GENERIC>(g, acc, smem, m0, n0, arow0, brow0, lane, tc.kbeg);
}

template <typename T, int ALAY, int BLAY, typename CF>
__global__ __launch_bounds__(CF::NT) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = Tr<T>::BK, BM = CF::BM, BN = CF::BN, NT = CF::NT, TM = CF::TM, TN = CF::TN;

  const TileCoord tc = tile_coord<CF>(g);
  const int m0 = tc.m0, n0 = tc.n0, kbeg = tc.kbeg, kend = tc.kend;

  const T* A = static_cast<const T*>(g.A);
  const T* B = static_cast<const T*>(g.B);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / CF::WN, wn = wave % CF::WN;
  const int arow0 = wm * TM * 32, brow0 = wn * TN * 32;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  typedef Stage<T, ALAY, BM, NT> SA;
  typedef Stage<T, BLAY, BN, NT> SB;
  SA sa;
  SB sb;
  sa.init(g.lda, m0, g.M);
  sb.init(g.ldb, n0, g.N);
  const T* pa = SA::base(A, g.lda, m0, kbeg);   // wave-uniform, advanced per K tile
  const T* pb = SB::base(B, g.ldb, n0, kbeg);
  const int nk = (kend - kbeg + BK - 1) / BK;
  const int ktail = (kend - kbeg) - (nk - 1) * BK;                    // valid k in the last tile (1..BK)
  const bool has_tail = ktail != BK;
  if (nk > 0) {
    if (nk == 1 && has_tail) {
      sa.load_tail(pa, ktail); sb.load_tail(pb, ktail);
      sa.store_tail(smem, ktail); sb.store_tail(smem + CF::A_BYTES, ktail);
    } else {
      sa.load(pa); sb.load(pb);
      sa.store(smem); sb.store(smem + CF::A_BYTES);
    }
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * CF::STAGE;
    char* nxt = smem + ((kt + 1) & 1) * CF::STAGE;
    const bool more = kt + 1 < nk;
    const bool tail_next = has_tail && kt + 2 == nk;
    if (more) {
      pa = SA::advance(pa, g.lda);
      pb = SB::advance(pb, g.ldb);
      if (tail_next) { sa.load_tail(pa, ktail); sb.load_tail(pb, ktail); }
      else { sa.load(pa); sb.load(pb); }
    }
    Mma<T, TM, TN>::tile(cur, cur + CF::A_BYTES, arow0, brow0, lane, acc);
    if (more) {
      if (tail_next) { sa.store_tail(nxt, ktail); sb.store_tail(nxt + CF::A_BYTES, ktail); }
      else { sa.store(nxt); sb.store(nxt + CF::A_BYTES); }
    }
    __syncthreads();
  }

  gemm_epilogue<T, CF, EK_GENERIC>(g, acc, smem, m0, n0, arow0, brow0, lane, kbeg);
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
static ProfSlot g_prof[11];   // 0-7: (dtype, layouts) of the 4-wave kernels; 8 / 9: wave-specialised 192 x 192 K,K / O,O kernels; 10: 128 x 192 K,K
static bool g_prof_on = false;
static int g_force_cfg = -1;   // tuning hook: force a geometry (0,1,2,3), 8 / 9: wave-specialised kernels never / always; -1 = heuristic

int gemm_forced_config() { return g_force_cfg; }

struct ProfToken { ProfSlot* ps; hipEvent_t e0; };
void* gemm_prof_begin(int slot, hipStream_t s) {
  if (!g_prof_on) return nullptr;
  ProfSlot* ps = &g_prof[slot < 11 ? slot : 0];
  hipEvent_t e0 = nullptr;
  if (ps->flops.size() >= 16384 || hipEventCreate(&e0) != hipSuccess) return nullptr;
  (void)hipEventRecord(e0, s);
  return new ProfToken{ps, e0};
}
void gemm_prof_end(void* token, double flops, hipStream_t s) {
  if (!token) return;
  ProfToken* t = static_cast<ProfToken*>(token);
  hipEvent_t e1 = nullptr;
  if (hipEventCreate(&e1) == hipSuccess) {
    (void)hipEventRecord(e1, s);
    t->ps->ev.push_back(t->e0);
    t->ps->ev.push_back(e1);
    t->ps->flops.push_back(flops);
  }
  delete t;
}
static int g_group = 0;        // tuning hook: M-tiles per locality group (0 = heuristic)

typedef Cfg<2, 2, 2, 2> Cfg128;
typedef Cfg<2, 4, 4, 2> Cfg256;
typedef Cfg<2, 2, 3, 2> Cfg192;    // 192x128: 20 % more FLOP per staged byte, 2 x 80 KB = the whole LDS of a CU
typedef Cfg<2, 2, 1, 1> Cfg64;     // 64x64 tiles for problems that leave most CUs idle at 128x128

template <typename T, int AL, int BL, typename CF>
static int launch(GemmArgs g, hipStream_t s) {
  HERO_ENSURE_LDS((&gemm_kernel<T, AL, BL, CF>), CF::LDS, "gemm_kernel");
  g.tiles_m = (g.M + CF::BM - 1) / CF::BM;
  g.tiles_n = (g.N + CF::BN - 1) / CF::BN;
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
  hipLaunchKernelGGL((gemm_kernel<T, AL, BL, CF>), dim3(grid), dim3(CF::NT), CF::LDS, s, g);
  if (ps) {
    (void)hipEventRecord(e1, s);
    ps->ev.push_back(e0);
    ps->ev.push_back(e1);
    ps->flops.push_back(2.0 * (double)g.M * (double)g.N * (double)g.K);
  }
  return check_launch("hero_gemm");
}

// Geometry choice (measured on MI355X, tools/gemm_bench.py): the 128x128 tile with two resident
// workgroups per CU is the best or within a few % of the best on every shape of the HERO step; the
// 256x256 tile wins ~5 % on the largest K-contiguous problems only.
static int pick_cfg(int M, int N, int split, bool k_contig) {
  if (g_force_cfg >= 0 && g_force_cfg < 8) return g_force_cfg;
  if (!k_contig || split != 1) return 0;
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  if (t128 <= 192) return 3;                                                // under one 128^2 tile per CU
  // a little over ONE round of the 512 resident slots (N = 768 at M = 12000: 564 tiles): 192x128 tiles
  // fit in a single round (378) - measured 12-18 % faster there, 5 % slower where 128^2 needs 3+ rounds
  if (t128 > 512 && t128 <= 700 && (long long)((M + 191) / 192) * ((N + 127) / 128) <= 512) return 1;
  // one 256x256 workgroup per CU: only worth it when the tiles fill whole rounds of the 256 CUs
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const long long rounds = (tiles + 255) / 256;
  return (tiles >= 256 && tiles * 10 >= rounds * 256 * 9) ? 2 : 0;
}

static int g_use_glds = 1;   // tuning hook

// explicit instantiations (hipcc does not emit the host stubs for kernels that are only reached
// through two levels of host-side templates)
#define HERO_GLDS_INST(T, CF)                                                       \
  template __global__ void gemm_glds_kernel<T, CF, EK_GENERIC>(GemmArgs);           \
  template __global__ void gemm_glds_kernel<T, CF, 0>(GemmArgs);                    \
  template __global__ void gemm_glds_kernel<T, CF, EK_BIAS>(GemmArgs);              \
  template __global__ void gemm_glds_kernel<T, CF, EK_BIAS | EK_RES | EK_DROP>(GemmArgs); \
  template __global__ void gemm_glds_kernel<T, CF, EK_BIAS | EK_GELU>(GemmArgs);    \
  template __global__ void gemm_glds_kernel<T, CF, EK_RES>(GemmArgs);               \
  template __global__ void gemm_glds_kernel<T, CF, EK_GELU_BWD>(GemmArgs);
HERO_GLDS_INST(bf16_t, Cfg128)
HERO_GLDS_INST(bf16_t, Cfg256)
HERO_GLDS_INST(bf16_t, Cfg64)
HERO_GLDS_INST(bf16_t, Cfg192)
template __global__ void gemm_glds_kernel<float, Cfg192, EK_GENERIC>(GemmArgs);
template __global__ void gemm_glds_kernel<float, Cfg64, EK_GENERIC>(GemmArgs);
template __global__ void gemm_glds_kernel<float, Cfg128, EK_GENERIC>(GemmArgs);
template __global__ void gemm_glds_kernel<float, Cfg256, EK_GENERIC>(GemmArgs);

template <typename T, typename CF, int EK>
static int launch_glds_ek(GemmArgs g, hipStream_t s) {
  HERO_ENSURE_LDS((&gemm_glds_kernel<T, CF, EK>), GldsRing<CF>::LDS, "gemm_glds_kernel");
  g.tiles_m = (g.M + CF::BM - 1) / CF::BM;
  g.tiles_n = (g.N + CF::BN - 1) / CF::BN;
  const int split = g.epi.split_k > 1 ? g.epi.split_k : 1;
  const int grid = g.tiles_m * g.tiles_n * split;
  ProfSlot* ps = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof_on) {
    ps = &g_prof[(sizeof(T) == 2 ? 4 : 0)];
    if (ps->flops.size() < 16384 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
      (void)hipEventRecord(e0, s);
    } else {
      ps = nullptr;
    }
  }
  hipLaunchKernelGGL((gemm_glds_kernel<T, CF, EK>), dim3(grid), dim3(CF::NT), GldsRing<CF>::LDS, s, g);
  if (ps) {
    (void)hipEventRecord(e1, s);
    ps->ev.push_back(e0);
    ps->ev.push_back(e1);
    ps->flops.push_back(2.0 * (double)g.M * (double)g.N * (double)g.K);
  }
  return check_launch("hero_gemm(glds)");
}

// epilogue specialisation: the hot-path combinations of the bf16 training step get their own
// instantiation, everything else (and all of f32) the generic one
template <typename T, typename CF>
static int launch_glds(const GemmArgs& g, hipStream_t s) {
  const HeroGemmEpilogue& e = g.epi;
  if constexpr (sizeof(T) == 2) if (!e.out_f32 && e.split_k <= 1) {
    const bool b = e.bias != nullptr, r = e.residual != nullptr, d = e.dropout.threshold16 != 0;
    if (e.act == HERO_ACT_NONE && b && !r && !d) return launch_glds_ek<T, CF, EK_BIAS>(g, s);
    if (e.act == HERO_ACT_NONE && b && r) return launch_glds_ek<T, CF, EK_BIAS | EK_RES | EK_DROP>(g, s);
    if ((e.act == HERO_ACT_GELU || e.act == HERO_ACT_GELU_DG) && b && !r && !d) return launch_glds_ek<T, CF, EK_BIAS | EK_GELU>(g, s);
    if (e.act == HERO_ACT_NONE && !b && !r && !d) return launch_glds_ek<T, CF, 0>(g, s);
    if (e.act == HERO_ACT_NONE && !b && r && !d) return launch_glds_ek<T, CF, EK_RES>(g, s);
    if ((e.act == HERO_ACT_GELU_BWD || e.act == HERO_ACT_MUL_AUX) && !b && !r && !d) return launch_glds_ek<T, CF, EK_GELU_BWD>(g, s);
  }
  return launch_glds_ek<T, CF, EK_GENERIC>(g, s);
}

static int launch_glds_tr(GemmArgs g, hipStream_t s) {
  typedef Cfg<2, 2, 2, 2> CF;
  g.tiles_m = (g.M + CF::BM - 1) / CF::BM;
  g.tiles_n = (g.N + CF::BN - 1) / CF::BN;
  const int split = g.epi.split_k > 1 ? g.epi.split_k : 1;
  const int grid = g.tiles_m * g.tiles_n * split;
  ProfSlot* ps = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof_on) {
    ps = &g_prof[7];
    if (ps->flops.size() < 16384 && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
      (void)hipEventRecord(e0, s);
    } else {
      ps = nullptr;
    }
  }
  HERO_ENSURE_LDS(&gemm_glds_tr_kernel, 65536, "gemm_glds_tr_kernel");
  hipLaunchKernelGGL(gemm_glds_tr_kernel, dim3(grid), dim3(256), 65536, s, g);
  if (ps) {
    (void)hipEventRecord(e1, s);
    ps->ev.push_back(e0);
    ps->ev.push_back(e1);
    ps->flops.push_back(2.0 * (double)g.M * (double)g.N * (double)g.K);
  }
  return check_launch("hero_gemm(glds_tr)");
}

template <typename T, int AL, int BL>
static int launch_cfg(const GemmArgs& g, int cfg, hipStream_t s) {
  if (AL == HERO_LAYOUT_O && BL == HERO_LAYOUT_O && sizeof(T) == 2 && g_use_glds && g.K > 0 && g.k_per_split % 64 == 0)
    return launch_glds_tr(g, s);
  if (AL == HERO_LAYOUT_K && BL == HERO_LAYOUT_K && g_use_glds && g.K > 0 && g.K % Tr<T>::BK == 0 &&
      g.k_per_split % Tr<T>::BK == 0) {
    if (cfg == 2) return launch_glds<T, Cfg256>(g, s);
    if (cfg == 3) return launch_glds<T, Cfg64>(g, s);
    if (cfg == 1) return launch_glds<T, Cfg192>(g, s);
    return launch_glds<T, Cfg128>(g, s);
  }
  if (cfg == 2) return launch<T, AL, BL, Cfg256>(g, s);
  return launch<T, AL, BL, Cfg128>(g, s);
}

template <typename T>
static int dispatch(const GemmArgs& g, int al, int bl, int cfg, hipStream_t s) {
  if (al == HERO_LAYOUT_K && bl == HERO_LAYOUT_K) return launch_cfg<T, HERO_LAYOUT_K, HERO_LAYOUT_K>(g, cfg, s);
  if (al == HERO_LAYOUT_K && bl == HERO_LAYOUT_O) return launch_cfg<T, HERO_LAYOUT_K, HERO_LAYOUT_O>(g, cfg, s);
  if (al == HERO_LAYOUT_O && bl == HERO_LAYOUT_O) return launch_cfg<T, HERO_LAYOUT_O, HERO_LAYOUT_O>(g, cfg, s);
  if (al == HERO_LAYOUT_O && bl == HERO_LAYOUT_K) return launch_cfg<T, HERO_LAYOUT_O, HERO_LAYOUT_K>(g, cfg, s);
  set_error("hero_gemm: bad layout (%d, %d)", al, bl);
  return HERO_ERR_ARG;
}

}  // namespace hero

using namespace hero;

// ---------------------------------------------------------------------------------------------
// Skinny fp32 GEMM: C[M <= 32, N] = A[M, K] B[N, K]^T (+ bias).  The loss head's fp32 Linear layers on the 32 query vectors
// (model/pretrain.py:128-166, model/encoder.py:426-485) are 32 x 768 x 768 / 32 x 1920 x 768: on the 64 x 64 MFMA tiles
// that is 12-30 workgroups walking 384 fp32 k-steps alone - 22 us per launch, three per micro-step.  Here: one wave per
// TWO output columns, every lane owns a 4-wide slice of the reduction per 256-k block (coalesced float4 loads of the weight
// rows and of the 32 activation rows, all independent -> one memory round trip), 64 fmaf chains per lane, a butterfly
// reduction per output; fixed order = bit-reproducible; ~96-240 workgroups.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes of a wave, valid in lane 63 (fixed order)
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xb1, 0xf>(v);      // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);      // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);     // row_ror:4
  v = dpp_add<0x128, 0xf>(v);     // row_ror:8   -> every lane of a 16-lane row holds the row's sum
  v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
  return v;
}

__global__ __launch_bounds__(256) void gemm_skinny_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                              const float* __restrict__ bias, int M, int N, int K, int lda, int ldb, int ldc) {
  // The activation rows go through the LDS (one cooperative copy per 1024-k chunk: a single memory round trip for the
  // workgroup, then ~100-cycle reads) - read straight from the L2 by every wave the three 256-k blocks of K = 768 were
  // three serial L2 round trips of 34 loads each (14 us per launch instead of 22; with the LDS copy ~6).
  extern __shared__ __attribute__((aligned(16))) char smem_skinny[];
  float* xs = reinterpret_cast<float*>(smem_skinny);
  constexpr int KC = 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * 2;
  const bool live = n0 < N, two = n0 + 1 < N;
  float acc0[32], acc1[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) acc0[m] = acc1[m] = 0.f;
  const float* b0 = B + (size_t)(live ? n0 : 0) * ldb;
  const float* b1 = B + (size_t)(two ? n0 + 1 : (live ? n0 : 0)) * ldb;
  for (int kc = 0; kc < K; kc += KC) {
    const int kw = min(KC, K - kc);                       // multiple of 4
    // this lane's weight slices of the chunk first (up to 4 x 2 float4): they come from HBM once per launch and their
    // latency is the kernel's critical path - in flight while the activation rows are copied
    float4 w0[4], w1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = min(lane * 4 + j * 256, kw - 4);
      w0[j] = *reinterpret_cast<const float4*>(b0 + kc + k);
      w1[j] = *reinterpret_cast<const float4*>(b1 + kc + k);
    }
    if (kc) __syncthreads();
    // Thread t copies the float4 at k = 4 t of every row: 16 loads in flight, then 16 LDS stores, twice.  Branch-free
    // (threads past the chunk's width load column 0 and store into a dummy slot behind the rows): with a branch around
    // each store hipcc sinks the load next to it and waits for every round trip - 24 serial L2 latencies for K = 768.
    {
      const int k = threadIdx.x * 4;
      const bool in = k < kw;
      const float* src = A + kc + (in ? k : 0);
      float* dst = in ? xs + k : xs + 32 * kw;
      const int dstride = in ? kw : 0;
#pragma unroll
      for (int m0 = 0; m0 < 32; m0 += 16) {
        float4 t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = *reinterpret_cast<const float4*>(src + (size_t)(m0 + j < M ? m0 + j : M - 1) * lda);
#pragma unroll
        for (int j = 0; j < 16; ++j) *reinterpret_cast<float4*>(dst + (m0 + j) * dstride) = t[j];
      }
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = lane * 4 + j * 256;
        if (k < kw) {
          // sixteen LDS reads issued as a block, then their 128 FMAs (left to itself hipcc waits for every read before the
          // FMAs that use it: 128 serial LDS latencies per chunk, 7 of the kernel's 11 us)
#pragma unroll
          for (int m0 = 0; m0 < 32; m0 += 16) {
            float4 xv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) xv[q] = *reinterpret_cast<const float4*>(xs + (m0 + q) * kw + k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float4 x = xv[q];
              acc0[m0 + q] = fmaf(x.w, w0[j].w, fmaf(x.z, w0[j].z, fmaf(x.y, w0[j].y, fmaf(x.x, w0[j].x, acc0[m0 + q]))));
              acc1[m0 + q] = fmaf(x.w, w1[j].w, fmaf(x.z, w1[j].z, fmaf(x.y, w1[j].y, fmaf(x.x, w1[j].x, acc1[m0 + q]))));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
  if (!live) return;
  // 64 wave-wide sums: DPP adds inside the VALU (quad swaps, row rotations, the two row broadcasts; the total ends up in
  // lane 63) - as __shfl_xor butterflies they were 384 ds_bpermute round trips, most of the kernel's 14 us.
  // lane m stores row m: pick its value without dynamic register indexing
  float v0 = 0.f, v1 = 0.f;
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_lane63(acc0[m])), 63));
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_lane63(acc1[m])), 63));
    v0 = lane == m ? s0 : v0;
    v1 = lane == m ? s1 : v1;
  }
  if (lane < M) {
    C[(size_t)lane * ldc + n0] = v0 + (bias ? bias[n0] : 0.f);
    if (two) C[(size_t)lane * ldc + n0 + 1] = v1 + (bias ? bias[n0 + 1] : 0.f);
  }
}

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
  HERO_REQUIRE(!(epi->act == HERO_ACT_GELU || epi->act == HERO_ACT_GELU_BWD || epi->act == HERO_ACT_RELU_BWD || epi->act == HERO_ACT_GELU_DG ||
                 epi->act == HERO_ACT_MUL_AUX) || epi->aux,
               "hero_gemm: activation %d needs aux", epi->act);
  HERO_REQUIRE(!epi->colsum || (epi->split_k <= 1 && !epi->out_f32), "hero_gemm: epilogue.colsum excludes split_k / out_f32");
  HERO_REQUIRE(epi->split_stride == 0 || (epi->split_k > 1 && epi->split_stride >= (long long)(M - 1) * ldc + N),
               "hero_gemm: split_stride needs split_k > 1 and slabs that hold [M, ldc]");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tiles_m = g.tiles_n = 0;
  // M-tiles per L2 locality group (tools/gemm_group_sweep.py): wide outputs like 16 (N = 3072: -4 %),
  // narrow ones 4 (N = 768: -2 %), 8 in between
  g.group = g_group ? g_group : (N >= 2560 ? 16 : (N <= 1024 ? 4 : 8));
  g.epi = *epi;
  const int bk = dtype == HERO_BF16 ? 64 : 32;
  const bool k_contig = a_layout == HERO_LAYOUT_K && b_layout == HERO_LAYOUT_K;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == HERO_F32 && k_contig && M <= 32 && K >= 256 && g_force_cfg == -1 && epi->act == HERO_ACT_NONE && !epi->residual && !epi->colsum &&
      !epi->out_f32 && epi->split_k <= 1 && epi->dropout.threshold16 == 0) {
    void* tok = gemm_prof_begin(3, s);
    const int lds = 32 * (K < 1024 ? K : 1024) * 4 + 16;       // + the dummy slot of the copy
    HERO_ENSURE_LDS(&gemm_skinny_f32_kernel, 32 * 1024 * 4 + 16, "gemm_skinny_f32_kernel");
    hipLaunchKernelGGL(gemm_skinny_f32_kernel, dim3((N + 7) / 8), dim3(256), lds, s, static_cast<const float*>(A), static_cast<const float*>(B),
                       static_cast<float*>(C), epi->bias, M, N, K, lda, ldb, ldc);
    gemm_prof_end(tok, 2.0 * (double)M * (double)N * (double)K, s);
    return check_launch("hero_gemm(skinny f32)");
  }
  if (dtype == HERO_BF16) {        // large problems: wave-specialised persistent 192 x 192 tiles (gemm_ws.hip)
    const int rc = gemm_ws_run(A, B, C, M, N, K, lda, ldb, ldc, a_layout, b_layout, *epi, g_force_cfg, s);
    if (rc != -1) return rc;
  }
  if (epi->split_k > 1) {
    HERO_REQUIRE(epi->out_f32 && epi->act == HERO_ACT_NONE && !epi->bias && !epi->residual && epi->dropout.threshold16 == 0,
                 "hero_gemm: split_k supports only a plain fp32 accumulate epilogue");
    const int ktiles = (K + bk - 1) / bk;
    int split = epi->split_k < ktiles ? epi->split_k : ktiles;
    if (split > 1) {
      const int per = (ktiles + split - 1) / split;
      split = (ktiles + per - 1) / per;
      if (split > 1) {
        if (epi->split_stride == 0 && epi->beta != 1.f) {  // atomics accumulate into beta * C (slabs are overwritten)
          hipLaunchKernelGGL(scale_f32_kernel, dim3(1024), dim3(256), 0, s, static_cast<float*>(C), M, N, ldc, epi->beta);
          const int rc = check_launch("hero_gemm(scale)");
          if (rc) return rc;
        }
        g.k_per_split = per * bk;
        g.epi.split_k = split;
        const int cfg = pick_cfg(M, N, split, k_contig);
        return dtype == HERO_BF16 ? dispatch<bf16_t>(g, a_layout, b_layout, cfg, s) : dispatch<float>(g, a_layout, b_layout, cfg, s);
      }
    }
  }
  g.k_per_split = K > 0 ? ((K + bk - 1) / bk) * bk : bk;
  g.epi.split_k = 1;
  const int cfg = pick_cfg(M, N, 1, k_contig);
  return dtype == HERO_BF16 ? dispatch<bf16_t>(g, a_layout, b_layout, cfg, s) : dispatch<float>(g, a_layout, b_layout, cfg, s);
}

extern "C" int hero_gemm_splits(int K, int split_k, int dtype) {
  const int bk = dtype == HERO_BF16 ? 64 : 32;
  const int ktiles = (K + bk - 1) / bk;
  int split = split_k < ktiles ? split_k : ktiles;
  if (split <= 1) return 1;
  const int per = (ktiles + split - 1) / split;
  return (ktiles + per - 1) / per;
}

// Tuning hook: force a tile geometry (0: 128x128, 1: 192x128, 2: 256x256, 3: 64x64 [1 and 3: direct-to-LDS path only], -1: heuristic).
extern "C" int hero_gemm_force_config(int cfg) {
  if (cfg >= 8 && cfg <= 14 && cfg != 11 && cfg != 12) {     // wave-specialised kernels: 8 never, 9 / 10 always 192 x 192 / 128 x 192, 13 / 14 the 64 x 128 / 64 x 192 geometries (small-M, long-K GEMMs)
    g_force_cfg = cfg;
    g_use_glds = 1;
    g_group = 0;
    return HERO_OK;
  }
  g_force_cfg = cfg >= 0 ? (cfg & 3) : -1;
  g_use_glds = cfg >= 0 ? !(cfg & 4) : 1;      // bit 2 set: register staging even for K,K operands
  g_group = cfg >= 0 ? (cfg >> 8) : 0;         // bits 8+: M-tiles per locality group (0 = heuristic)
  return HERO_OK;
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
  HERO_REQUIRE(slot >= 0 && slot < 11 && total_ms && total_flops && launches, "hero_prof_read: bad arguments");
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
