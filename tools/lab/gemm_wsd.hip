// Wave-specialised persistent bf16 GEMM, K,K operands, with a DEFERRED epilogue (round 4).
//
//   C[M,N] bf16 = A[M,K] B[N,K]^T (+ one of the six hot-path epilogues of gemm_ws.hip)
//
// Same tiles, ring, LDS images, loader-wave DMA stream and compute-wave main loop as gemm_ws_kernel (see the top of
// gemm_ws.hip).  What changes is where a finished tile goes.  In gemm_ws_kernel all eight waves run the epilogue between
// two main loops: three passes of (stage 64 rows in LDS, barrier, read rows + arithmetic + 16-byte stores, barrier), and
// because the 256 persistent workgroups walk equal tiles in lock step, all of them store at the same time - the stores
// back up into the issuing waves (tools/lab/trace_ws.py: 300-390 cycles per store instruction) while the HBM sits idle
// during the main loops.  That epilogue is 6 800 (bias) to 17 000 (bias + GELU + saved pre-activation) cycles of a
// 27 000-40 000-cycle K = 768 tile, and the ring slot it borrows costs the next tile another ~2 700 cycles.
//
// Here the loader waves - which own 256 registers each and use two dozen - PARK the tile: per 64-row pass the compute
// waves stage their accumulators in the ring slot that was read last (as before), the loader waves copy the raw fp32
// values into registers (18 x 8 floats per lane for a 192 x 192 tile), and after the last pass the compute waves go
// straight into the next tile's main loop.  The loader waves then drain the parked tile DURING that main loop, two
// (row, 8-column) items per ring step: bias / GELU / dropout / residual / gelu' on the parked values, 16-byte stores.
// The stores of a launch are spread over its main loops, the arithmetic runs beside the MFMAs of the partner wave, and
// the hand-off costs the compute waves ~3 x (staging + LDS read + two barriers).
//
// vmcnt bookkeeping of a loader wave (VMEM operations of one wave complete in order on gfx9, loads and stores alike -
// the compiler's own waitcnt insertion relies on it): per drain step the wave issues, in program order,
//     DMA(stage u+2) x PW | wait | barrier | R(next items) x NR | arithmetic(current items) | St(current items) x NST
// and the wait of the NEXT step leaves exactly St x NST + DMA x PW outstanding: everything older - the stage the barrier
// is about to publish and the residual / saved pre-activation of the items about to be processed - has landed, while the
// stores get a whole ring step to complete before anything waits for them.
#include "gemm_ws_common.h"

namespace hero {
namespace ws {

template <int TM, int TN, int EK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wsd_kernel(WsArgs g) {
  typedef Geo<TM, TN> G;
  constexpr int C8 = G::C8, RPP = G::RPP, PASSES = G::PASSES, BN = G::BN;
  constexpr int IPP = RPP * C8 / 256;          // (row, 8-column) items per loader thread per pass
  constexpr int NITEM = IPP * PASSES;          // ... per tile: 18 (192 x 192), 12 (128 x 192)
  constexpr int IPS = 2;                       // items drained per ring step
  constexpr bool HAS_PRE = (EK & (EK_RES | EK_GELU_BWD)) != 0;
  constexpr int SPI = (EK & EK_GELU) ? 2 : 1;  // stores per item
  constexpr int NST = IPS * SPI;
  constexpr int DSTEPS = NITEM / IPS;           // ring steps a drain takes: the launcher requires K / 64 >= DSTEPS
  constexpr int PW = G::PW;
  static_assert(RPP * C8 % 256 == 0 && NITEM % IPS == 0 && DSTEPS >= 1, "item mapping");
  constexpr bool SPLIT = (RPP == 64 && PASSES == TM);
  auto tile_row = [](int p, int row) { return SPLIT ? (row >> 5) * (TM * 32) + p * 32 + (row & 31) : p * RPP + row; };

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves: DMA stream + drain
    const int w = wave - 4;
    const int ltid = w * 64 + lane;
    const HeroGemmEpilogue& e = g.epi;
    DropCtx drop(e.dropout);
    const bool use_drop = (EK & EK_DROP) && drop.on();
    bf16_t* Cb = static_cast<bf16_t*>(g.C);
    const bf16_t* R = (EK & EK_RES) ? static_cast<const bf16_t*>(e.residual) : nullptr;
    bf16_t* X = static_cast<bf16_t*>(e.aux);
    float* bsp = reinterpret_cast<float*>(smem + SPARE_OFF);          // the parked tile's bias columns
    Loader<G, false> ld(g, smem, wg, nwg, w, lane);
    ld.issue_always();
    ld.issue_always();
    wait_vm<PW>();
    __builtin_amdgcn_s_barrier();                                     // B(-1): stage 0 landed
    unsigned slot = 0;
    float park[NITEM][8];
    uint4 pre[2][IPS];
    Item pic = {0, 0, 0, 0};
    bool pending = false;
    int item_no = 0;

    for (int cit = wg;; cit += nwg, ++item_no) {
      // One more trip than the workgroup has tiles: the last parked tile is drained behind DSTEPS dummy ring steps (zero-range
      // DMA, barriers among the loader waves only - the compute waves have left), i.e. by the same straight-line code.
      const bool have = cit < g.nwork;
      Item ic = {0, 0, 0, 0};
      if (have) ic = item_coord<G>(g, cit);
      const int nk = have ? ic.nk : DSTEPS;                           // ic.nk >= DSTEPS (launcher)
      int t = 0;
      WS_T(item_no, 0, wave, lane);
      if (pending) {
        // per-tile constants of the PARKED tile (tile-relative descriptors: the matrices may exceed 2^31 bytes)
        const size_t torg = (size_t)pic.m0 * g.ldc + pic.n0;
        const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(Cb + torg, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((EK & (EK_GELU | EK_GELU_BWD)) ? X + torg : Cb + torg, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>((EK & EK_RES) ? R + torg : Cb + torg), 0, 0x7fffffff, 0x00020000);
        auto load_pre = [&](int ci) __attribute__((always_inline)) -> uint4 {
          const int p = ci / IPP, q = ci % IPP;
          int lt = ltid;
          asm volatile("" : "+v"(lt));        // opaque: hoisted out of the tile loop these addresses get spilled (scratch is VMEM)
          const int id = q * 256 + lt, prow = id / C8, c8 = id - prow * C8;
          const int gm = pic.m0 + tile_row(p, prow);
          const int gnc = min(pic.n0 + c8 * 8, g.N - 8);
          const unsigned off = ((unsigned)(min(gm, g.M - 1) - pic.m0) * (unsigned)g.ldc + (unsigned)(gnc - pic.n0)) * 2u;
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128((EK & EK_RES) ? rsr : rsx, off, 0, 0);
          return uint4{v[0], v[1], v[2], v[3]};
        };
        auto drain_item = [&](int ci, const uint4& pr) __attribute__((always_inline)) {
          const int p = ci / IPP, q = ci % IPP;
          int lt = ltid;
          asm volatile("" : "+v"(lt));
          const int id = q * 256 + lt, prow = id / C8, c8 = id - prow * C8;
          const int trow = tile_row(p, prow);
          const int gm = pic.m0 + trow, gn = pic.n0 + c8 * 8;
          const bool ok = gm < g.M && gn < g.N;
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = park[ci][k];
          if (EK & EK_BIAS) {
            const int bc = min(c8 * 8, g.N - 8 - pic.n0);
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bsp + bc);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(bsp + bc + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
          }
          uint4 u = {0u, 0u, 0u, 0u};
          if (EK & EK_GELU) {
            if (e.act == HERO_ACT_GELU_DG) {               // uniform: save gelu'(v) instead of v
              float dg[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) gelu_both<bf16_t>(v[k], v[k], dg[k]);
              u.x = f2bf_pk(dg[0], dg[1]); u.y = f2bf_pk(dg[2], dg[3]); u.z = f2bf_pk(dg[4], dg[5]); u.w = f2bf_pk(dg[6], dg[7]);
            } else {
              u.x = f2bf_pk(v[0], v[1]); u.y = f2bf_pk(v[2], v[3]); u.z = f2bf_pk(v[4], v[5]); u.w = f2bf_pk(v[6], v[7]);
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] = gelu_fwd<bf16_t>(v[k]);
            }
          }
          float pv[8];
          if (HAS_PRE) {
            const uint32_t w4[4] = {pr.x, pr.y, pr.z, pr.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              pv[2 * k] = __uint_as_float(w4[k] << 16);
              pv[2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u);
            }
          }
          if (EK & EK_GELU_BWD) {
            if (e.act == HERO_ACT_MUL_AUX) {
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] *= pv[k];
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) v[k] *= gelu_grad<bf16_t>(pv[k]);
            }
          }
          if (use_drop) {
            const uint64_t grp = ((uint64_t)gm * (uint64_t)g.N + (uint64_t)gn) >> 2;
            const float4 m0 = drop.mask4(grp), m1 = drop.mask4(grp + 1);
            v[0] *= m0.x; v[1] *= m0.y; v[2] *= m0.z; v[3] *= m0.w;
            v[4] *= m1.x; v[5] *= m1.y; v[6] *= m1.z; v[7] *= m1.w;
          }
          if (EK & EK_RES) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += pv[k];
          }
          // branch-free stores: a masked-off lane gets an offset past the descriptor's range and the hardware drops it
          const unsigned vo = ok ? (unsigned)(trow * g.ldc + c8 * 8) * 2u : 0xffffffffu;
          if (EK & EK_GELU) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{u.x, u.y, u.z, u.w}, rsx, vo, 0, HERO_WS_STORE_AUX2);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7])},
                                                 rsc, vo, 0, HERO_WS_STORE_AUX);
        };
        // DSTEPS ring steps, each followed by the arithmetic and stores of IPS parked items (straight-line: every path
        // issues the same VMEM operations, so the compiler's own waits agree with the explicit ones)
#pragma unroll
        for (int c = 0; c < NITEM; c += IPS) {
          ld.issue_always();                                            // stage u+2
          if (c == 0) wait_vm<PW>(); else wait_vm<PW + NST>();          // stage u+1 and R(current items) have landed
          __builtin_amdgcn_s_barrier();                                 // B(u)
          if (t + 1 < nk) { slot += G::STAGE; if (slot == NS * G::STAGE) slot = 0; }
          ++t;
          if (HAS_PRE && c + IPS < NITEM) {
#pragma unroll
            for (int k = 0; k < IPS; ++k) pre[((c / IPS) + 1) & 1][k] = load_pre(c + IPS + k);
          }
#pragma unroll
          for (int k = 0; k < IPS; ++k) drain_item(c + k, pre[(c / IPS) & 1][k]);
        }
        pending = false;
        WS_T(item_no, 2, wave, lane);
      }
      for (; t < nk; ++t) {
        ld.issue_always();
        wait_vm<PW>();
        __builtin_amdgcn_s_barrier();                                   // B(u)
        if (t + 1 < nk) { slot += G::STAGE; if (slot == NS * G::STAGE) slot = 0; }
      }
      if (!have) break;
      WS_T(item_no, 1, wave, lane);
      // ---------------------------------------------------------------- hand-off: park the tile
      if (EK & EK_BIAS) {
        if (ltid < BN) bsp[ltid] = e.bias[min(ic.n0 + ltid, g.N - 1)];
        wait_lds();                                                     // visible behind the H1 barrier
      }
      const char* st = smem + slot;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        __builtin_amdgcn_s_barrier();                                   // H1: pass p is staged
#pragma unroll
        for (int q = 0; q < IPP; ++q) {
          int lt = ltid;
          asm volatile("" : "+v"(lt));
          const int id = q * 256 + lt, prow = id / C8, c8 = id - prow * C8, x = prow & 7;
          const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(st + prow * G::ROWB + (((2 * c8) ^ x) << 4));
          const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(st + prow * G::ROWB + (((2 * c8 + 1) ^ x) << 4));
#pragma unroll
          for (int k = 0; k < 4; ++k) { park[p * IPP + q][k] = v0[k]; park[p * IPP + q][4 + k] = v1[k]; }
        }
        wait_lds();
        __builtin_amdgcn_s_barrier();                                   // H2: the slot may be restaged / refilled
      }
      pic = ic;
      pending = true;
      if (HAS_PRE) {
        const size_t torg = (size_t)pic.m0 * g.ldc + pic.n0;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>((EK & EK_RES) ? R + torg : X + torg), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int k = 0; k < IPS; ++k) {
          int lt = ltid;
          asm volatile("" : "+v"(lt));
          const int id = k * 256 + lt, prow = id / C8, c8 = id - prow * C8;
          const int gm = pic.m0 + tile_row(0, prow);
          const int gnc = min(pic.n0 + c8 * 8, g.N - 8);
          const unsigned off = ((unsigned)(min(gm, g.M - 1) - pic.m0) * (unsigned)g.ldc + (unsigned)(gnc - pic.n0)) * 2u;
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
          pre[0][k] = uint4{v[0], v[1], v[2], v[3]};
        }
      }
      WS_T(item_no, 3, wave, lane);
      slot += G::STAGE; if (slot == NS * G::STAGE) slot = 0;
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wm = wave >> 1, wn = wave & 1;
  const int arow0 = wm * TM * 32, brow0 = wn * TN * 32;
  unsigned ao[TM], bo[TN];                                            // per-lane LDS offsets inside a stage (slice 0)
  {
    const int r = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int ra = arow0 + i * 32 + r; ao[i] = ra * 128 + ((kg ^ swz_k(ra)) << 4); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int rb = brow0 + j * 32 + r; bo[j] = G::A_BYTES + rb * 128 + ((kg ^ swz_k(rb)) << 4); }
  }
  bf16x8_t a0[TM], b0[TN], a1[TM], b1[TN];
  auto ldf = [&](bf16x8_t (&a)[TM], bf16x8_t (&b)[TN], const char* st, int ks) {       // order a[0], b[..], a[1..]: see gemm_ws_kernel
    a[0] = *reinterpret_cast<const bf16x8_t*>(st + (ao[0] ^ (ks << 5)));
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(st + (bo[j] ^ (ks << 5)));
#pragma unroll
    for (int i = 1; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(st + (ao[i] ^ (ks << 5)));
  };
  f32x16_t acc[TM][TN];
  auto mma = [&](const bf16x8_t (&a)[TM], const bf16x8_t (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // D^T: lane <-> output row
  };

  __builtin_amdgcn_s_setprio(2);
  __builtin_amdgcn_s_barrier();                                       // B(-1)
  unsigned curo = 0;
  if (wg < g.nwork) ldf(a0, b0, smem, 0);
  int item_no = 0;
  for (int cit = wg; cit < g.nwork; cit += nwg, ++item_no) {
    const Item ic = item_coord<G>(g, cit);
    WS_T(item_no, 0, wave, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const bool more_items = cit + nwg < g.nwork;
    unsigned last = curo;
    for (int t = 0; t < ic.nk; ++t) {
      const char* cur = smem + curo;
      last = curo;
      curo += G::STAGE;
      if (curo == NS * G::STAGE) curo = 0;
      const char* nxt = smem + curo;
      constexpr int NRD = TM + TN;
      ldf(a1, b1, cur, 1);
      mma(a0, b0);
      WS_INTERLEAVE(TM * TN, NRD);
      __builtin_amdgcn_sched_barrier(0);
      ldf(a0, b0, cur, 2);
      mma(a1, b1);
      WS_INTERLEAVE(TM * TN, NRD);
      __builtin_amdgcn_sched_barrier(0);
      ldf(a1, b1, cur, 3);
      mma(a0, b0);
      WS_INTERLEAVE(TM * TN, NRD);
      __builtin_amdgcn_sched_barrier(0);
      wait_lds();
      __builtin_amdgcn_s_barrier();                                   // B(u): done reading `cur`, stage u+1 landed
      __builtin_amdgcn_sched_barrier(0);
      ldf(a0, b0, nxt, 0);            // unconditional: behind an item's last step it reads the next item's landed first
      mma(a1, b1);                    // stage (read again after the hand-off) or stale LDS, never used
      WS_INTERLEAVE(TM * TN, NRD);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    WS_T(item_no, 1, wave, lane);
    // ------------------------------------------------------------------ hand-off: stage the accumulators, pass by pass
    {
      char* st = smem + last;
      const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
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
        wait_lds();
        __builtin_amdgcn_s_barrier();                    // H1: the pass is staged
        __builtin_amdgcn_s_barrier();                    // H2: the loader waves hold it
      }
    }
    WS_T(item_no, 3, wave, lane);
    __builtin_amdgcn_s_setprio(2);
    if (more_items) ldf(a0, b0, smem + curo, 0);
  }
}

#define HERO_WSD_INST(TM, TN)                                                                  \
  template __global__ void gemm_wsd_kernel<TM, TN, 0>(WsArgs);                                 \
  template __global__ void gemm_wsd_kernel<TM, TN, EK_BIAS>(WsArgs);                           \
  template __global__ void gemm_wsd_kernel<TM, TN, EK_BIAS | EK_RES | EK_DROP>(WsArgs);        \
  template __global__ void gemm_wsd_kernel<TM, TN, EK_BIAS | EK_GELU>(WsArgs);                 \
  template __global__ void gemm_wsd_kernel<TM, TN, EK_RES>(WsArgs);                            \
  template __global__ void gemm_wsd_kernel<TM, TN, EK_GELU_BWD>(WsArgs);
HERO_WSD_INST(3, 3)
HERO_WSD_INST(2, 3)

static int wsd_num_cus() {
  static int n = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : 256;
  }();
  return n;
}

template <int TM, int TN, int EK>
static int launch_d(const WsArgs& g, int slot, hipStream_t s) {
  typedef Geo<TM, TN> G;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wsd_kernel<TM, TN, EK>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    attr_set = true;
  }
  const int grid = g.nwork < wsd_num_cus() ? g.nwork : wsd_num_cus();
  void* tok = gemm_prof_begin(slot, s);
  hipLaunchKernelGGL((gemm_wsd_kernel<TM, TN, EK>), dim3(grid), dim3(512), G::LDS, s, g);
  gemm_prof_end(tok, 2.0 * (double)g.M * (double)g.N * (double)g.K, s);
  return check_launch("hero_gemm(wsd)");
}

// The deferred-epilogue flavour of launch_kk (gemm_ws.hip): same problem envelope; the gelu' epilogue WITH column sums
// stays on the in-line kernel (its bias gradient rides on the batched wgrad instead, functional.FfnBlockFn).
template <int TM, int TN>
int launch_kk_deferred(const WsArgs& g, hipStream_t s) {
  const HeroGemmEpilogue& e = g.epi;
  const bool b = e.bias != nullptr, r = e.residual != nullptr, d = e.dropout.threshold16 != 0;
  const int slot = TM == 3 ? 8 : 10;
  if (e.colsum != nullptr) return -1;
  // a parked tile is drained in NITEM / 2 ring steps of the next tile (9 for 192 x 192, 6 for 128 x 192)
  typedef Geo<TM, TN> G;
  if (g.K / 64 < (G::RPP * G::C8 / 256) * G::PASSES / 2) return -1;
  if (e.act == HERO_ACT_NONE && b && !r && !d) return launch_d<TM, TN, EK_BIAS>(g, slot, s);
  if (e.act == HERO_ACT_NONE && b && r) return launch_d<TM, TN, EK_BIAS | EK_RES | EK_DROP>(g, slot, s);
  if ((e.act == HERO_ACT_GELU || e.act == HERO_ACT_GELU_DG) && b && !r && !d) return launch_d<TM, TN, EK_BIAS | EK_GELU>(g, slot, s);
  if (e.act == HERO_ACT_NONE && !b && !r && !d) return launch_d<TM, TN, 0>(g, slot, s);
  if (e.act == HERO_ACT_NONE && !b && r && !d) return launch_d<TM, TN, EK_RES>(g, slot, s);
  if ((e.act == HERO_ACT_GELU_BWD || e.act == HERO_ACT_MUL_AUX) && !b && !r && !d) return launch_d<TM, TN, EK_GELU_BWD>(g, slot, s);
  return -1;
}
template int launch_kk_deferred<3, 3>(const WsArgs&, hipStream_t);
template int launch_kk_deferred<2, 3>(const WsArgs&, hipStream_t);

}  // namespace ws
}  // namespace hero

#ifdef HERO_WS_TRACE
extern "C" int hero_wsd_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hero::ws::g_ws_trace), sizeof(unsigned long long) * 4 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif
