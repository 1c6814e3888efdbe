// Shared pieces of the wave-specialised persistent GEMM family (gemm_ws.hip: in-line epilogue, wgrad flavours;
// gemm_wsd.hip: the K,K kernel whose epilogue is drained by the loader waves during the next tile's main loop).
// Geometry, work-item order, LDS images and the loader-wave DMA stream are described at the top of gemm_ws.hip.
#pragma once
#include <vector>

#include "gemm_args.h"

namespace hero {
namespace ws {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

constexpr int NS = 3;                 // ring stages of the 192 x 192 / 128 x 192 geometries (Geo::NSG: per geometry)
constexpr int SPARE_OFF = 144 * 1024; // 16 KiB behind the largest ring: column-sum fold

// Cache policy of the epilogue stores (buffer-store aux bits; 2 = nt: streaming, no allocation priority in the L2).
// Measured on the bench step (same box): both K,K outputs nt 52.8 us per launch / 6.96 ms per step against 54.7 / 7.05
// with the default policy - the output of a tile is not read again by this kernel, and the panels the other CUs are
// re-reading stay in the L2.
#ifndef HERO_WS_STORE_AUX
#define HERO_WS_STORE_AUX 2          // the main output of a K,K tile
#endif
#ifndef HERO_WS_STORE_AUX2
#define HERO_WS_STORE_AUX2 2         // the saved pre-activation (read again only in the backward pass)
#endif
#ifndef HERO_WS_LOAD_AUX_A
#define HERO_WS_LOAD_AUX_A 0         // cache policy of the direct-to-LDS operand loads (lab)
#endif
#ifndef HERO_WS_LOAD_AUX_B
#define HERO_WS_LOAD_AUX_B 0
#endif
#ifndef HERO_WS_STORE_DW
#define HERO_WS_STORE_DW 0           // dW tiles of the batched wgrad (read again by the optimiser)
#endif
#define HERO_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Scheduling pattern of a region that holds NM MFMAs and ND LDS reads (ND <= 2 NM): MFMA, PER reads, MFMA, PER reads, ...
// (sched_group_barrier masks: 0x008 MFMA, 0x100 DS read).  HERO_WS_BLOCKED restores the round-2 order for A/B runs.
template <int NM, int ND, int PER>
__device__ __forceinline__ void ws_interleave() {
#ifndef HERO_WS_BLOCKED
  if constexpr (NM > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (ND >= PER) __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
    else if constexpr (ND > 0) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
    ws_interleave<NM - 1, (ND >= PER ? ND - PER : 0), PER>();
  }
#else
  __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
#endif
}
#define WS_INTERLEAVE(NM, ND) ws_interleave<(NM), (ND), ((ND) > (NM) ? 2 : 1)>()

#ifdef HERO_WS_TRACE
// timeline probe (tools/lab/trace_ws.py): s_memtime stamps of the first four items of workgroup 0, per wave
static __device__ unsigned long long g_ws_trace[4 * 16 * 8];   // one per translation unit
#define WS_T(item_no, ev, wave, lane)                                                                   \
  do {                                                                                                  \
    if (blockIdx.x == 0 && (lane) == 0 && (item_no) >= 0 && (item_no) < 4)                              \
      g_ws_trace[((item_no) * 16 + (ev)) * 8 + (wave)] = __builtin_readcyclecounter();                  \
  } while (0)
#else
#define WS_T(item_no, ev, wave, lane) do { } while (0)
#endif

struct WsArgs {
  const void* A;
  const void* B;
  void* C;
  int M, N, K, lda, ldb, ldc;       // output M x N, reduction K
  int tiles_m, tiles_n, group, nsplit, k_per_split, nwork;
  HeroGemmEpilogue epi;
};

template <int TM_, int TN_> struct Geo {
  static constexpr int TM = TM_, TN = TN_;
  static constexpr int BM = 64 * TM, BN = 64 * TN;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int PA = BM / 32, PB = BN / 32, PW = PA + PB;      // 1-KiB pieces per loader wave per stage
  static constexpr int ROWB = BN * 4;                                   // fp32 staging row
  static constexpr int RPP = (STAGE / ROWB >= 64 && BM % 64 == 0) ? 64 : 32;   // rows per epilogue pass
  static constexpr int PASSES = BM / RPP;
  static constexpr int C8 = BN / 8, RPI = 512 / C8, ITERS = (RPP + RPI - 1) / RPI;
  // ring depth: what fits under the spare region, at most 6.  The 64-row geometries (small-M GEMMs) move 24-32 KB per step
  // and need the deeper ring to keep as many bytes in flight as the large tiles do with three stages.
  static constexpr int NSG = SPARE_OFF / STAGE > 6 ? 6 : SPARE_OFF / STAGE;
  static constexpr int LDS = SPARE_OFF + 16384;
  static_assert(NSG >= 3 && NSG * STAGE <= SPARE_OFF && RPP * ROWB <= STAGE && RPI * BN * 4 <= 16384 && (NSG - 2) * PW < 64, "LDS budget");
  static_assert(TM_ < 2 || NSG == NS, "the large geometries keep the three-stage ring");
};

struct Item { int m0, n0, kbeg, nk; };

template <typename G>
__device__ __forceinline__ Item item_coord(const WsArgs& g, int item) {
  const int ntile = g.tiles_m * g.tiles_n;
  const int split = item / ntile;
  const int tile = item - split * ntile;
  const int per_group = g.group * g.tiles_n;
  const int group = tile / per_group;
  const int first_m = group * g.group;
  const int gsz = min(g.tiles_m - first_m, g.group);
  const int in_group = tile - group * per_group;
  Item it;
  it.m0 = (first_m + in_group % gsz) * G::BM;
  it.n0 = (in_group / gsz) * G::BN;
  it.kbeg = split * g.k_per_split;
  it.nk = (min(g.K, it.kbeg + g.k_per_split) - it.kbeg + 63) >> 6;
  return it;
}

__device__ __forceinline__ int swz_k(int row) { return (row ^ (row >> 3)) & 7; }
// O,O image: chunk swizzle of reduction row k for a tile row of RB bytes
template <int RB> __device__ __forceinline__ int swz_o(int k) { return RB % 256 == 0 ? 4 * (k & 3) : 4 * ((k >> 1) & 1); }

// ------------------------------------------------------------------------------------------------
// loader waves
// ------------------------------------------------------------------------------------------------
template <typename G, bool TR>
struct Loader {
  const WsArgs& g;
  char* smem;
  int w, lane, nwg;
  int item, ik;           // item / stage being issued next
  Item ic;
  unsigned fill;
  unsigned goa[G::PA], gob[G::PB];
  const char* pa;         // stage base of the A / B panels (uniform)
  const char* pb;
  unsigned ra_left, rb_left;   // bytes from the stage base to the end of the operand (O,O bounds)

  __device__ __forceinline__ Loader(const WsArgs& g_, char* smem_, int wg, int nwg_, int w_, int lane_)
      : g(g_), smem(smem_), w(w_), lane(lane_), nwg(nwg_), item(wg), ik(0), fill(0) {
    if (item < g.nwork) setup();
  }
  __device__ __forceinline__ void setup() {
    ic = item_coord<G>(g, item);
    const bf16_t* A = static_cast<const bf16_t*>(g.A);
    const bf16_t* B = static_cast<const bf16_t*>(g.B);
    if (!TR) {
#pragma unroll
      for (int i = 0; i < G::PA; ++i) {
        const int r = (w * G::PA + i) * 8 + (lane >> 3);
        goa[i] = (unsigned)(min(ic.m0 + r, g.M - 1) - ic.m0) * (unsigned)g.lda * 2u + (((lane & 7) ^ swz_k(r)) << 4);
      }
#pragma unroll
      for (int i = 0; i < G::PB; ++i) {
        const int r = (w * G::PB + i) * 8 + (lane >> 3);
        gob[i] = (unsigned)(min(ic.n0 + r, g.N - 1) - ic.n0) * (unsigned)g.ldb * 2u + (((lane & 7) ^ swz_k(r)) << 4);
      }
      pa = reinterpret_cast<const char*>(A + (size_t)ic.m0 * g.lda + ic.kbeg);
      pb = reinterpret_cast<const char*>(B + (size_t)ic.n0 * g.ldb + ic.kbeg);
      ra_left = rb_left = 0x7fffffffu;
    } else {
      constexpr int CA = G::BM / 8, CB = G::BN / 8;      // 16-B chunks per tile row
#pragma unroll
      for (int i = 0; i < G::PA; ++i) {
        const int id = (w * G::PA + i) * 64 + lane, row = id / CA, c = (id % CA) ^ swz_o<G::BM * 2>(row);
        goa[i] = (unsigned)row * (unsigned)g.lda * 2u + (c << 4);
      }
#pragma unroll
      for (int i = 0; i < G::PB; ++i) {
        const int id = (w * G::PB + i) * 64 + lane, row = id / CB, c = (id % CB) ^ swz_o<G::BN * 2>(row);
        gob[i] = (unsigned)row * (unsigned)g.ldb * 2u + (c << 4);
      }
      // A is [K, lda] with the tile's M columns at m0; B is [K, ldb] with the N columns at n0
      pa = reinterpret_cast<const char*>(A + (size_t)ic.kbeg * g.lda + ic.m0);
      pb = reinterpret_cast<const char*>(B + (size_t)ic.kbeg * g.ldb + ic.n0);
      ra_left = (unsigned)(((size_t)(g.K - ic.kbeg) * g.lda - ic.m0) * 2);
      rb_left = (unsigned)(((size_t)(g.K - ic.kbeg) * g.ldb - ic.n0) * 2);
    }
  }
  // gemm_wsd_kernel: issue PW pieces in EVERY step, so that the wave's vmcnt arithmetic is the same on every path.  Behind
  // the end of the item stream the descriptors have a zero range: the loads fetch nothing, write zeros into a ring slot
  // nobody reads again, and count like the real ones.
  __device__ __forceinline__ void issue_always() {
    if (item >= g.nwork) { ra_left = rb_left = 0u; ic.nk = 0x7fffffff; }
    issue_body();
  }
  // issue the next stage of the item stream (false: the stream has ended)
  __device__ __forceinline__ bool issue() {
    if (item >= g.nwork) return false;
    issue_body();
    return true;
  }
  __device__ __forceinline__ void issue_body() {
    char* buf = smem + fill;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pa), 0, ra_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, rb_left, 0x00020000);
#pragma unroll
    for (int i = 0; i < G::PA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, HERO_LDS_PTR(buf + (w * G::PA + i) * 1024), 16, goa[i], 0, 0, HERO_WS_LOAD_AUX_A);
#pragma unroll
    for (int i = 0; i < G::PB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, HERO_LDS_PTR(buf + G::A_BYTES + (w * G::PB + i) * 1024), 16, gob[i], 0, 0, HERO_WS_LOAD_AUX_B);
    fill += G::STAGE;
    if (fill == G::NSG * G::STAGE) fill = 0;
    if (++ik == ic.nk) {
      item += nwg;
      ik = 0;
      if (item < g.nwork) setup();
    } else if (!TR) {
      pa += 128;
      pb += 128;
    } else {
      const unsigned sa = 64u * (unsigned)g.lda * 2u, sb = 64u * (unsigned)g.ldb * 2u;
      pa += sa; pb += sb;
      ra_left = ra_left > sa ? ra_left - sa : 0u;
      rb_left = rb_left > sb ? rb_left - sb : 0u;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// K,K epilogue: run by all 8 waves; the compute waves additionally stage their accumulators
// ------------------------------------------------------------------------------------------------
template <typename G, int EK, bool COMPUTE>
__device__ __forceinline__ void epilogue_rows(const WsArgs& g, const Item& ic, char* smem, unsigned slot, f32x16_t (*acc)[G::TN], int wave,
                                              int lane, int trace_item = -1) {
  constexpr int TM = G::TM, TN = G::TN, BN = G::BN, RPP = G::RPP, C8 = G::C8, RPI = G::RPI, ITERS = G::ITERS;
  const HeroGemmEpilogue& e = g.epi;
  char* st = smem + slot;
  // Everything the epilogue derives from the thread index is recomputed per tile from an opaque copy: hoisted out of the
  // item loop these values stay live across the main loop, where 144 accumulators + 48 fragment registers leave no room,
  // and get spilled to scratch (a reload = one memory round trip at the start of every epilogue).
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  lane = tid & 63;
  wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c8 = tid % C8, r0 = tid / C8;
  const bool active = r0 < RPI;
  const int gn = ic.n0 + c8 * 8;
  const bool col_ok = active && gn < g.N;
  const int gnc = min(gn, g.N - 8);
  bf16_t* Cb = static_cast<bf16_t*>(g.C);
  const bf16_t* R = (EK & EK_RES) ? static_cast<const bf16_t*>(e.residual) : nullptr;
  bf16_t* X = static_cast<bf16_t*>(e.aux);
  // tile-relative store descriptors (offsets inside a tile stay far below 2^31 bytes whatever the size of C)
  const size_t torg = (size_t)ic.m0 * g.ldc + ic.n0;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(Cb + torg, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((EK & EK_GELU) ? X + torg : Cb + torg, 0, 0x7fffffff, 0x00020000);
  DropCtx drop(e.dropout);
  const bool use_drop = (EK & EK_DROP) && drop.on();
  float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (EK & EK_BIAS) {
    const float4 b0 = *reinterpret_cast<const float4*>(e.bias + gnc);
    const float4 b1 = *reinterpret_cast<const float4*>(e.bias + gnc + 4);
    bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w;
    bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
  }
  const bool do_csum = (EK & EK_GELU_BWD) && e.colsum != nullptr;
  const bool save_dg = (EK & EK_GELU) && e.act == HERO_ACT_GELU_DG, mul_aux = (EK & EK_GELU_BWD) && e.act == HERO_ACT_MUL_AUX;
  const bool relu = (EK & EK_GELU) && e.act == HERO_ACT_RELU;      // uniform: the activation-with-saved-value instantiations serve ReLU too
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  // A pass stages 64 rows.  With three passes over a 192-row tile the two 32-row blocks of a pass are taken from the
  // two wave rows (block p of each), so that all four compute waves stage 12 fragments per pass instead of two waves
  // staging 24 (the staging of a pass was 2100 cycles of a 5300-cycle pass, tools/lab/trace_ws.py).
  constexpr bool SPLIT = (RPP == 64 && G::PASSES == TM);
  auto tile_row = [](int p, int row) { return SPLIT ? (row >> 5) * (TM * 32) + p * 32 + (row & 31) : p * RPP + row; };

#pragma unroll
  for (int p = 0; p < G::PASSES; ++p) {
    // residual / saved pre-activation of this pass: fetched before the accumulators are staged
    uint4 pre[ITERS];
    unsigned off[ITERS];
    bool ok[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = r0 + it * RPI;
      const int gm = ic.m0 + tile_row(p, row);
      ok[it] = col_ok && row < RPP && gm < g.M;
      // tile-relative (like the stores): the matrix itself may hold more than 2^32 elements (config 5: 1.5 M rows x 3072)
      off[it] = (unsigned)(min(gm, g.M - 1) - ic.m0) * (unsigned)g.ldc + (unsigned)(gnc - ic.n0);
      if (EK & EK_RES) pre[it] = *reinterpret_cast<const uint4*>(R + torg + off[it]);
      if (EK & EK_GELU_BWD) pre[it] = *reinterpret_cast<const uint4*>(X + torg + off[it]);
    }
    WS_T(trace_item, 2 + 4 * p, wave, lane);
    if constexpr (COMPUTE) {
#pragma unroll
      for (int b = 0; b < RPP / 32; ++b) {
        const int blk = SPLIT ? b * TM + p : p * (RPP / 32) + b;   // 32-row block of the tile
        if (wm == blk / TM) {
          const int i = blk % TM;                     // compile-time after unrolling
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int chunk = (wn * TN * 32 + j * 32 + 8 * q + 4 * half) >> 2;
              const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
              *reinterpret_cast<f32x4_t*>(st + (32 * b + l31) * G::ROWB + ((chunk ^ (l31 & 7)) << 4)) = v;
            }
        }
      }
    }
    wait_lds();
    WS_T(trace_item, 3 + 4 * p, wave, lane);
    __builtin_amdgcn_s_barrier();                    // E1: the pass is staged
    WS_T(trace_item, 4 + 4 * p, wave, lane);
    // All iterations are computed first (branch-free: inactive threads and rows past the pass read a clamped row and
    // store nothing), THEN the stores are issued back to back.  With a store inside each iteration the compiler put
    // an s_waitcnt vmcnt(0) in front of the next iteration's arithmetic (its registers were the store's data), i.e.
    // every iteration waited for the previous store's round trip to HBM: 4 serialised round trips per pass.
    uint4 outv[ITERS], auxv[(EK & EK_GELU) ? ITERS : 1];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = min(r0 + it * RPI, RPP - 1);
      {
        const int x = row & 7;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(st + row * G::ROWB + (((2 * c8) ^ x) << 4));
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(st + row * G::ROWB + (((2 * c8 + 1) ^ x) << 4));
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bias[k];
        if (EK & EK_GELU) {
          uint4 u;
          if (save_dg) {                                // uniform (HERO_ACT_GELU_DG): the derivative is saved, not the pre-activation
            float dg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) gelu_both<bf16_t>(v[k], v[k], dg[k]);
            u.x = f2bf_pk(dg[0], dg[1]); u.y = f2bf_pk(dg[2], dg[3]); u.z = f2bf_pk(dg[4], dg[5]); u.w = f2bf_pk(dg[6], dg[7]);
          } else if (relu) {                            // HERO_ACT_RELU: aux <- relu(acc + bias), the value before the residual
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            u.x = f2bf_pk(v[0], v[1]); u.y = f2bf_pk(v[2], v[3]); u.z = f2bf_pk(v[4], v[5]); u.w = f2bf_pk(v[6], v[7]);
          } else {
            u.x = f2bf_pk(v[0], v[1]); u.y = f2bf_pk(v[2], v[3]); u.z = f2bf_pk(v[4], v[5]); u.w = f2bf_pk(v[6], v[7]);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = gelu_fwd<bf16_t>(v[k]);
          }
          auxv[(EK & EK_GELU) ? it : 0] = u;
        }
        float pv[8];                                  // residual / saved pre-activation as fp32
        if (EK & (EK_RES | EK_GELU_BWD)) {
          const uint32_t w4[4] = {pre[it].x, pre[it].y, pre[it].z, pre[it].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pv[2 * k] = __uint_as_float(w4[k] << 16);
            pv[2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u);
          }
        }
        if (EK & EK_GELU_BWD) {
          if (mul_aux) {                                // uniform (HERO_ACT_MUL_AUX): aux already holds gelu'
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= pv[k];
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= gelu_grad<bf16_t>(pv[k]);
          }
        }
        if (use_drop) {
          const int gm = ic.m0 + tile_row(p, row);
          const uint64_t grp = ((uint64_t)gm * (uint64_t)g.N + (uint64_t)gn) >> 2;
          const float4 m0 = drop.mask4(grp), m1 = drop.mask4(grp + 1);
          v[0] *= m0.x; v[1] *= m0.y; v[2] *= m0.z; v[3] *= m0.w;
          v[4] *= m1.x; v[5] *= m1.y; v[6] *= m1.z; v[7] *= m1.w;
        }
        if (EK & EK_RES) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += pv[k];
        }
        if (do_csum && ok[it]) {
#pragma unroll
          for (int k = 0; k < 8; ++k) cs[k] += v[k];
        }
        uint4 o;
        o.x = f2bf_pk(v[0], v[1]); o.y = f2bf_pk(v[2], v[3]); o.z = f2bf_pk(v[4], v[5]); o.w = f2bf_pk(v[6], v[7]);
        outv[it] = o;
      }
    }
    // Branch-free stores: a masked-off lane gets an offset past the descriptor's range and the hardware drops it.
    // (Behind `if (ok)` every store sat in its own basic block, and each block re-waited vmcnt(0) for the bias /
    // residual loads of the tile start - which by then also meant the previous block's store.)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned vo = ok[it] ? (unsigned)(tile_row(p, r0 + it * RPI) * g.ldc + c8 * 8) * 2u : 0xffffffffu;
      if (EK & EK_GELU) {
        const uint4 u = auxv[(EK & EK_GELU) ? it : 0];
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{u.x, u.y, u.z, u.w}, rsx, vo, 0, HERO_WS_STORE_AUX2);
      }
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{outv[it].x, outv[it].y, outv[it].z, outv[it].w}, rsc, vo, 0, HERO_WS_STORE_AUX);
    }
    wait_lds();
    WS_T(trace_item, 5 + 4 * p, wave, lane);
    __builtin_amdgcn_s_barrier();                    // E2: the slot may be restaged / refilled
  }
  WS_T(trace_item, 14, wave, lane);
  if (EK & EK_GELU_BWD) {                            // uniform across the workgroup (kernel argument)
    if (e.colsum != nullptr) {
      float* sp = reinterpret_cast<float*>(smem + SPARE_OFF);
      if (active) {
        *reinterpret_cast<f32x4_t*>(sp + r0 * BN + c8 * 8) = f32x4_t{cs[0], cs[1], cs[2], cs[3]};
        *reinterpret_cast<f32x4_t*>(sp + r0 * BN + c8 * 8 + 4) = f32x4_t{cs[4], cs[5], cs[6], cs[7]};
      }
      wait_lds();
      __builtin_amdgcn_s_barrier();                  // E3
      if (tid < BN && ic.n0 + tid < g.N) {
        float t = 0.f;
#pragma unroll 4
        for (int k = 0; k < RPI; ++k) t += sp[k * BN + tid];
        if (e.colsum_partial) {
          // deterministic: this tile's sums go to row (m0 / 64) of the [ceil(M / 64), N] partial table, zeros to the rows of
          // the tile's other 64-row blocks - whatever the tile height, the table's column sums are the result
          float* pr = e.colsum + (size_t)(ic.m0 >> 6) * g.N + ic.n0 + tid;
          const int nb = min(G::BM >> 6, ((g.M + 63) >> 6) - (ic.m0 >> 6));
          pr[0] = t;
          for (int b = 1; b < nb; ++b) pr[(size_t)b * g.N] = 0.f;
        } else {
          atomicAdd(e.colsum + ic.n0 + tid, t);
        }
      }
      // the next writer of the spare region is the next tile's fold, >= one step barrier away
    }
  }
}


}  // namespace ws
}  // namespace hero
