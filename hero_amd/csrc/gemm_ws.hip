// Wave-specialised persistent bf16 GEMM for gfx950 (MI355X): the large-M shapes of the HERO step.
//
//   K,K operands (forward x W^T, dgrad dY (W^T)^T):   C[M,N] bf16 = A[M,K] B[N,K]^T (+ fused epilogue)
//   O,O operands (wgrad dW += dY^T X, rows reduced):    C[M,N] fp32 += A[K,M]^T B[K,N], reduction split
//
// One 512-thread workgroup per CU, persistent over a static list of work items (output tile x
// reduction split), tile (64 TM) x (64 TN) x 64:
//   waves 4-7  LOADERS: issue the direct-to-LDS loads (buffer_load ... lds, 16 B / lane, 1 KiB per
//              instruction) of the item stream into a 3-stage ring, two stages ahead, counted vmcnt.  The
//              per-lane offsets are fixed for an item, the k advance moves the descriptor base (SALU),
//              M0 carries the LDS destination: no VALU per load, and the compute waves' instruction
//              streams never carry a VMEM issue (60-180 cycles each beside MFMAs).
//   waves 0-3  COMPUTE (2 x 2, TM x TN MFMA 32x32x16 tiles each): fragments of the next 16-k slice are read
//              from LDS into a second register set while the MFMAs of the current slice run.
//   One raw s_barrier per 64-k step hands the ring over (stage t+1 landed / stage t free).  The stream
//   runs across item boundaries, so the next tile's first two stages land during the epilogue.
// The 192 x 192 tile is the shape of the step: M = 12000 rows and N = 768 k give 63 x 4 k tiles = 252 k
// workgroups on 256 CUs (98 % fill for N = 768, 2304, 3072; 256 x 256 tiles fill 55 % of one round at N = 768).
//
// (Tried and dropped, round 2: the same tile with 4 waves that load AND compute, 32-k stages, two workgroups per
// CU so that one's epilogue overlaps the other's main loop - 69 vs 55 us at N = 768, K = 3072 and 73 vs 67 us at
// N = 3072, K = 768.  A 6-stage ring (120 KiB in flight) ran at the same 0.72 us per 32-k step as the 3-stage one:
// not load latency but the VMEM issue slots inside the MFMA stream, which is what the loader waves remove.)
//
// (Tried and dropped, round 2: a stream-K cut of the K,K item list - equal k-step ranges, fp32 hand-over of the split
// tiles through a workspace with device-coherent (sc1) stores / loads, item order by ownership so that the workgroups
// still sweep the list together - to spread the epilogues over the main loops of other CUs: 129 vs 66 us at N = 3072.
// The write-through hand-over of 256 x 147 KB costs 21 us, the coherent read-back 5 us, and the de-synchronised
// epilogue itself was only ~13 % shorter: it is latency-structured (3 passes x staging + 2 barriers + row pass),
// not purely HBM-bound.  Agent-scope fences instead of sc1 accesses were worse: each one writes back / drops the L2.)
//
// Epilogue, K,K: accumulators -> LDS (fp32, the ring slot that was read last, 64 rows per pass, 16-byte
// chunks XOR-swizzled by row) -> all 8 waves apply bias / GELU / dropout / residual / gelu' on full rows
// and store 16 B per lane.  The MFMA operands are swapped (D^T = B A^T) so that a lane holds 4
// consecutive columns of one row: the LDS writes are ds_write_b128.  O,O: fp32 atomics straight from the
// (unswapped) accumulator layout: 32 consecutive columns per half-wave.
//
// LDS images: K,K  [rows][128 B] per operand, 16-B chunk c of row r at chunk c ^ ((r ^ r>>3) & 7)
//             O,O  [64 k][BM * 2 B], chunk c of row k at c ^ swz(k): the four k-rows a 32-lane half of a
//                  ds_read_b64_tr_b16 touches land in four distinct 64-B quarters of the bank row
// (the swizzle is applied on the SOURCE address of the DMA and on the read address).
// Reduction tail of the O,O flavour (rows % 64 != 0): the loads of rows past the matrix are
// out-of-range for the buffer descriptor and deliver zeros.
#include <stdlib.h>
#include <type_traits>

#include "gemm_ws_common.h"

namespace hero {
namespace ws {

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <int TM, int TN, bool TR, int EK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_ws_kernel(WsArgs g) {
  typedef Geo<TM, TN> G;
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
    // ------------------------------------------------------------------ loader waves
    // The ring holds NSG stages: NSG - 1 are issued ahead, the wave then always waits until all but the NSG - 2 youngest
    // have landed (VMEM operations of a wave complete in order), i.e. until the stage the compute waves read next is in.
    Loader<G, TR> ld(g, smem, wg, nwg, wave - 4, lane);
    constexpr int AHEAD = (G::NSG - 2) * G::PW;
    bool more = ld.issue();
#pragma unroll
    for (int k = 0; k < G::NSG - 2; ++k) more = ld.issue();
    if (more) wait_vm<AHEAD>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                                     // B(-1): stage 0 landed
    unsigned slot = 0;
    int item_no = 0;
    for (int cit = wg; cit < g.nwork; cit += nwg, ++item_no) {
      const Item ic = item_coord<G>(g, cit);
      WS_T(item_no, 0, wave, lane);
      for (int t = 0; t < ic.nk; ++t) {
        if (ld.issue()) wait_vm<AHEAD>(); else wait_vm<0>();          // stage u + NSG - 1 issued, stage u + 1 landed
        __builtin_amdgcn_s_barrier();                                 // B(u)
        if (t + 1 < ic.nk) { slot += G::STAGE; if (slot == G::NSG * G::STAGE) slot = 0; }
      }
      WS_T(item_no, 1, wave, lane);
#ifndef HERO_WS_NOEPI      // lab ablation (tools/lab/gemm_ceiling.sh): the main loops alone, outputs discarded - timing only
      if constexpr (!TR) epilogue_rows<G, EK, false>(g, ic, smem, slot, nullptr, wave, lane, item_no);
#endif
      slot += G::STAGE; if (slot == G::NSG * G::STAGE) slot = 0;
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wm = wave >> 1, wn = wave & 1;
  const int arow0 = wm * TM * 32, brow0 = wn * TN * 32;
  unsigned ao[TM], bo[TN];                                            // per-lane LDS offsets inside a stage (slice 0)
  if (!TR) {
    const int r = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int ra = arow0 + i * 32 + r; ao[i] = ra * 128 + ((kg ^ swz_k(ra)) << 4); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int rb = brow0 + j * 32 + r; bo[j] = G::A_BYTES + rb * 128 + ((kg ^ swz_k(rb)) << 4); }
  } else {
    const int p = lane & 15, gq = (lane >> 4) & 1, kg = lane >> 5;
    const int krow = kg * 8 + (p >> 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int col = arow0 + i * 32 + gq * 16 + 4 * (p & 3);
      ao[i] = krow * (G::BM * 2) + ((((col >> 3) ^ swz_o<G::BM * 2>(krow)) << 4) | ((col & 7) * 2));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = brow0 + j * 32 + gq * 16 + 4 * (p & 3);
      bo[j] = G::A_BYTES + krow * (G::BN * 2) + ((((col >> 3) ^ swz_o<G::BN * 2>(krow)) << 4) | ((col & 7) * 2));
    }
  }
  bf16x8_t a0[TM], b0[TN], a1[TM], b1[TN];
  // Read order a[0], b[0..], a[1..]: the MFMAs of the NEXT slice run (i outer, j inner), the reads are spread over the
  // MFMAs of the current slice in this order, so every fragment is requested >= 7 MFMAs (224 cycles) before its first
  // use (a[0..], b[0..] order: 5 MFMAs for b[0] - less than the LDS latency beside the DMA writes).
  auto ldf = [&](bf16x8_t (&a)[TM], bf16x8_t (&b)[TN], const char* st, int ks) {
    typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
    auto ra = [&](int i) {
      if (!TR) {
        a[i] = *reinterpret_cast<const bf16x8_t*>(st + (ao[i] ^ (ks << 5)));
      } else {
        const char* q = st + ao[i] + ks * 16 * (G::BM * 2);
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (G::BM * 2)));
        a[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    auto rb = [&](int j) {
      if (!TR) {
        b[j] = *reinterpret_cast<const bf16x8_t*>(st + (bo[j] ^ (ks << 5)));
      } else {
        const char* q = st + bo[j] + ks * 16 * (G::BN * 2);
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (G::BN * 2)));
        b[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    if (!TR) {                                      // K,K: the MFMA is (b[j], a[i]) but the loop order is the same
      ra(0);
#pragma unroll
      for (int j = 0; j < TN; ++j) rb(j);
#pragma unroll
      for (int i = 1; i < TM; ++i) ra(i);
    } else {
      ra(0);
#pragma unroll
      for (int j = 0; j < TN; ++j) rb(j);
#pragma unroll
      for (int i = 1; i < TM; ++i) ra(i);
    }
  };
  f32x16_t acc[TM][TN];
  auto mma = [&](const bf16x8_t (&a)[TM], const bf16x8_t (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!TR) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // D^T: lane <-> output row
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
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
      if (curo == G::NSG * G::STAGE) curo = 0;
      const char* nxt = smem + curo;
      // one scheduling region per 16-k slice: the fragment reads of the NEXT slice are spread between the MFMAs of the
      // current one (round 3; as a block in front of them they cost MFMA-idle issue time, see gemm_wsb_kernel)
      constexpr int NRD = TR ? 2 * (TM + TN) : TM + TN;
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
      mma(a1, b1);                    // stage (K,K: read again after the epilogue) or stale LDS, never used
      WS_INTERLEAVE(TM * TN, NRD);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!TR) {
      __builtin_amdgcn_s_setprio(0);
      WS_T(item_no, 1, wave, lane);
#ifndef HERO_WS_NOEPI
      epilogue_rows<G, EK, true>(g, ic, smem, last, acc, wave, lane, item_no);
#else
      {                                     // every accumulator element stays live (or the compiler deletes MFMAs); never true
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 1.2345e-30f) static_cast<bf16_t*>(g.C)[lane] = (bf16_t)1;
      }
#endif
      __builtin_amdgcn_s_setprio(2);
      if (more_items) ldf(a0, b0, smem + curo, 0);
    } else {
      // fp32 atomics from the accumulator layout: 32 consecutive columns per half-wave, two rows per instruction
      float* C = static_cast<float*>(g.C);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int gn = ic.n0 + brow0 + j * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int gm = ic.m0 + arow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gm < g.M && gn < g.N) atomicAdd(C + (size_t)gm * g.ldc + gn, acc[i][j][r]);
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Grouped wgrad with a stream-K decomposition: up to 4 problems dW_p += dY_p^T X_p that reduce over the SAME rows
// (the four weight gradients of a transformer layer: 64 + 64 + 16 + 48 = 192 tiles of 192 x 192 over 12000 rows) in
// ONE launch.  The (tile, k-step) space of the whole group is cut into nwg equal contiguous ranges, one per
// workgroup (256 x 140 tile-steps instead of four launches that each quantise to whole tiles x splits and each
// end in their own merge tail).  A range covers the end of one tile and the beginning of the next (sometimes
// a whole tile in between): each piece ("segment") is accumulated like an item of the kernel above - the loader
// waves stream straight across the segment boundaries - and added to C with fp32 atomics (every tile receives
// 1-3 partial sums).  Same tile geometry, ring, wave roles and O,O images as gemm_ws_kernel<3, 3, true>.
// ------------------------------------------------------------------------------------------------
struct WsgProb {
  const bf16_t* A;      // dY  [K, lda], M columns from the pointer on
  const bf16_t* B;      // X   [K, ldb], N columns
  float* C;             // dW  [M, ldc]
  int M, N, lda, ldb, ldc, tiles_n, tile0;     // tile0: index of this problem's first tile in the group order
};
struct WsgArgs {
  WsgProb p[4];
  int nprob, K, ksteps, total_tiles;
};
struct Seg { int prob, m0, n0, k0, nk; };

template <typename G>
__device__ __forceinline__ Seg seg_at(const WsgArgs& g, int pos, int end) {
  const int tile = pos / g.ksteps, k0 = pos - tile * g.ksteps;
  int pi = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q < g.nprob && tile >= g.p[q].tile0) pi = q;
  const int t = tile - g.p[pi].tile0;
  Seg s;
  s.prob = pi;
  s.m0 = (t / g.p[pi].tiles_n) * G::BM;
  s.n0 = (t % g.p[pi].tiles_n) * G::BN;
  s.k0 = k0;
  s.nk = min(g.ksteps - k0, end - pos);
  return s;
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wsg_kernel(WsgArgs g) {
  typedef Geo<3, 3> G;
  constexpr int TM = 3, TN = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const long long total = (long long)g.total_tiles * g.ksteps;
  const int start = (int)(total * wg / nwg), end = (int)(total * (wg + 1) / nwg);
  if (start >= end) return;                                          // uniform for the workgroup

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int w = wave - 4;
    constexpr int CA = G::BM / 8, CB = G::BN / 8;
    int pos = start;                 // next stage to issue belongs to the segment that contains `pos`
    Seg sg = seg_at<G>(g, pos, end);
    int ik = 0;
    unsigned goa[G::PA], gob[G::PB];
    const char* pa = nullptr;
    const char* pb = nullptr;
    unsigned ra_left = 0, rb_left = 0, sa = 0, sb = 0, fill = 0;
    auto setup = [&]() {
      const WsgProb& P = g.p[sg.prob];
#pragma unroll
      for (int i = 0; i < G::PA; ++i) {
        const int id = (w * G::PA + i) * 64 + lane, row = id / CA, c = (id % CA) ^ swz_o<G::BM * 2>(row);
        goa[i] = (unsigned)row * (unsigned)P.lda * 2u + (c << 4);
      }
#pragma unroll
      for (int i = 0; i < G::PB; ++i) {
        const int id = (w * G::PB + i) * 64 + lane, row = id / CB, c = (id % CB) ^ swz_o<G::BN * 2>(row);
        gob[i] = (unsigned)row * (unsigned)P.ldb * 2u + (c << 4);
      }
      const int kb = sg.k0 * 64;
      pa = reinterpret_cast<const char*>(P.A + (size_t)kb * P.lda + sg.m0);
      pb = reinterpret_cast<const char*>(P.B + (size_t)kb * P.ldb + sg.n0);
      ra_left = (unsigned)(((size_t)(g.K - kb) * P.lda - sg.m0) * 2);
      rb_left = (unsigned)(((size_t)(g.K - kb) * P.ldb - sg.n0) * 2);
      sa = 64u * (unsigned)P.lda * 2u;
      sb = 64u * (unsigned)P.ldb * 2u;
    };
    setup();
    auto issue = [&]() -> bool {
      if (pos >= end) return false;
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
      if (fill == NS * G::STAGE) fill = 0;
      ++pos;
      if (++ik == sg.nk) {
        ik = 0;
        if (pos < end) { sg = seg_at<G>(g, pos, end); setup(); }
      } else {
        pa += sa; pb += sb;
        ra_left = ra_left > sa ? ra_left - sa : 0u;
        rb_left = rb_left > sb ? rb_left - sb : 0u;
      }
      return true;
    };
    issue();
    const bool second = issue();
    if (second) wait_vm<G::PW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                                     // B(-1)
    for (int u = start; u < end; ++u) {
      if (issue()) wait_vm<G::PW>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();                                   // B(u)
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wm = wave >> 1, wn = wave & 1;
  const int arow0 = wm * TM * 32, brow0 = wn * TN * 32;
  unsigned ao[TM], bo[TN];
  {
    const int p = lane & 15, gq = (lane >> 4) & 1, kg = lane >> 5;
    const int krow = kg * 8 + (p >> 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int col = arow0 + i * 32 + gq * 16 + 4 * (p & 3);
      ao[i] = krow * (G::BM * 2) + ((((col >> 3) ^ swz_o<G::BM * 2>(krow)) << 4) | ((col & 7) * 2));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = brow0 + j * 32 + gq * 16 + 4 * (p & 3);
      bo[j] = G::A_BYTES + krow * (G::BN * 2) + ((((col >> 3) ^ swz_o<G::BN * 2>(krow)) << 4) | ((col & 7) * 2));
    }
  }
  typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
  bf16x8_t a0[TM], b0[TN], a1[TM], b1[TN];
  auto ldf = [&](bf16x8_t (&a)[TM], bf16x8_t (&b)[TN], const char* st, int ks) {       // order a[0], b[..], a[1..]: see gemm_ws_kernel
    auto ra = [&](int i) {
      const char* q = st + ao[i] + ks * 16 * (G::BM * 2);
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (G::BM * 2)));
      a[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    ra(0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const char* q = st + bo[j] + ks * 16 * (G::BN * 2);
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (G::BN * 2)));
      b[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 1; i < TM; ++i) ra(i);
  };
  f32x16_t acc[TM][TN];
  auto mma = [&](const bf16x8_t (&a)[TM], const bf16x8_t (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  __builtin_amdgcn_s_setprio(2);
  __builtin_amdgcn_s_barrier();                                       // B(-1)
  unsigned curo = 0;
  ldf(a0, b0, smem, 0);
  [[maybe_unused]] int seg_no = 0;
  for (int pos = start; pos < end; ++seg_no) {
    const Seg sg = seg_at<G>(g, pos, end);
    WS_T(seg_no, 0, wave, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int t = 0; t < sg.nk; ++t) {
      const char* cur = smem + curo;
      curo += G::STAGE;
      if (curo == NS * G::STAGE) curo = 0;
      const char* nxt = smem + curo;
      ldf(a1, b1, cur, 1);
      mma(a0, b0);
      WS_INTERLEAVE(TM * TN, 2 * (TM + TN));
      __builtin_amdgcn_sched_barrier(0);
      ldf(a0, b0, cur, 2);
      mma(a1, b1);
      WS_INTERLEAVE(TM * TN, 2 * (TM + TN));
      __builtin_amdgcn_sched_barrier(0);
      ldf(a1, b1, cur, 3);
      mma(a0, b0);
      WS_INTERLEAVE(TM * TN, 2 * (TM + TN));
      __builtin_amdgcn_sched_barrier(0);
      wait_lds();
      __builtin_amdgcn_s_barrier();                                   // B(u)
      __builtin_amdgcn_sched_barrier(0);
      ldf(a0, b0, nxt, 0);            // unconditional (stale LDS behind the range's last step, never used)
      mma(a1, b1);
      WS_INTERLEAVE(TM * TN, 2 * (TM + TN));
      __builtin_amdgcn_sched_barrier(0);
    }
    pos += sg.nk;
    WS_T(seg_no, 1, wave, lane);
#ifdef HERO_WS_TRACE
    if (blockIdx.x == 0 && lane == 0 && seg_no < 4) g_ws_trace[(seg_no * 16 + 3) * 8 + wave] = (unsigned long long)sg.nk;
#endif
    const WsgProb& P = g.p[sg.prob];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int gn = sg.n0 + brow0 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = sg.m0 + arow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < P.M && gn < P.N) atomicAdd(P.C + (size_t)gm * P.ldc + gn, acc[i][j][r]);
        }
      }
    WS_T(seg_no, 14, wave, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// Batched wgrad, whole tiles (round 3): up to 32 problems dW_p += dY_p^T X_p that reduce over the SAME rows - the weight
// gradients of ALL the BertLayers of an encoder (6 x (64 + 64 + 16 + 48) = 1152 tiles of 192 x 192 over 12000 rows =
// 4.5 rounds of 256 CUs) - in ONE launch, scheduled by a host-built plan (hero_wgrad_batch_plan):
//   * full rounds: every workgroup owns WHOLE tiles (all k-steps), so there is no merge: dW += acc is a plain fp32
//     read-add-write of the tile, bit-reproducible.  All workgroups start their tiles at k = 0 together and the 32 tiles
//     of an XCD in a round are a compact patch of ONE problem (8 x 4 tiles: 12 distinct operand panels instead of 64), so
//     concurrent tiles share their panels in the XCD's L2 (the stream-K ranges of gemm_wsg_kernel start at staggered k
//     offsets and share nothing: 765 MB of fabric traffic per launch against 320 MB algorithmic);
//   * tail round (tiles % 256 != 0): the remaining tiles are cut into S k-slices so that tiles x S fills the chip; the
//     slices of a tile add to dW with fp32 atomics IN SLICE ORDER (a per-tile flag: slice s waits until slice s - 1 has
//     drained its atomics), which keeps the sum bit-reproducible.
// Same ring / wave roles / O,O LDS images as gemm_ws_kernel<3, 3, true>; the MFMA operands are swapped (acc = dW^T
// fragments: lane <-> output row) so that the accumulators are staged through LDS with ds_write_b128 and all eight
// waves update full rows of dW with 16-byte accesses.
// ------------------------------------------------------------------------------------------------
struct WsbProb {
  const bf16_t* A;      // dY [K, lda], M columns from the pointer on
  const bf16_t* B;      // X  [K, ldb], N columns
  float* C;             // dW [M, ldc]
  float* colsum;        // optional [M]: += column sums of dY (bias gradient), from the tiles with n0 == 0
  int M, N, lda, ldb, ldc, pad_;
};
struct WsbItem { int prob, m0, n0, k0, nk, order, nslices, flag; };     // nk == 0: the slot is idle in this round
struct WsbArgs {
  const WsbItem* items;   // [rounds][nwg]
  int* flags;             // slice-order flags of the tail tiles (zero between launches)
  int rounds, K, nprob, pad_;
  WsbProb p[HERO_WGRAD_BATCH_MAX];
};

// want_cs (loader waves of a tile with n0 == 0 and a bias gradient): the compute waves with wn == 0 have left the column sums
// of the tile's dY panel in the spare LDS region ([BM] floats); loader waves 0-2 apply them behind the first pass barrier -
// inside the slice-order window, so the bias gradient is as reproducible as dW.
template <typename G, bool COMPUTE>
__device__ __forceinline__ void epilogue_acc(const WsbProb& P, const WsbItem& it, int* flags, char* smem, unsigned slot,
                                             f32x16_t (*acc)[G::TN], int wave, int lane, bool want_cs = false) {
  constexpr int TM = G::TM, TN = G::TN, BN = G::BN, RPP = G::RPP, C8 = G::C8, RPI = G::RPI, ITERS = G::ITERS;
  char* st = smem + slot;
  int tid = threadIdx.x;                             // opaque copy: see epilogue_rows
  asm volatile("" : "+v"(tid));
  lane = tid & 63;
  wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c8 = tid % C8, r0 = tid / C8;
  const int gn = it.n0 + c8 * 8;
  const bool col_ok = r0 < RPI && gn < P.N;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  const bool plain = it.nslices == 1;                                 // uniform
  constexpr bool SPLIT = (RPP == 64 && G::PASSES == TM);
  auto tile_row = [](int p, int row) { return SPLIT ? (row >> 5) * (TM * 32) + p * 32 + (row & 31) : p * RPP + row; };
  // dW through a buffer descriptor: masked-off lanes get an out-of-range offset (loads return 0, stores are dropped),
  // so every access of a pass is issued back to back in straight-line code (a branch per store makes hipcc wait for
  // the previous store's round trip in every iteration: 58 us instead of 7 per tile)
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(P.C, 0, 0x7ffffff0, 0x00020000);
  if (!plain && it.order > 0) {
    // earlier slices first: their atomics have drained (vmcnt(0) + barrier) before the flag moves
    if (lane == 0)
      while (__hip_atomic_load(flags + it.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != it.order) __builtin_amdgcn_s_sleep(8);
  }
#pragma unroll
  for (int p = 0; p < G::PASSES; ++p) {
    u32x4_t old0[ITERS], old1[ITERS];
    unsigned voff[ITERS];                            // byte offsets inside dW (< 2^31: checked by the launcher)
    if (plain) {
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        const int row = r0 + i * RPI;
        const int gm = it.m0 + tile_row(p, row);
        const bool ok = col_ok && row < RPP && gm < P.M;
        voff[i] = ok ? ((unsigned)gm * (unsigned)P.ldc + (unsigned)gn) * 4u : 0xfffffff0u;
        // the tile's old values: in flight while the accumulators are staged
#if defined(HERO_WSB_EPI_NOLOAD) || defined(HERO_WSB_EPI_NOMEM)
        old0[i] = u32x4_t{0u, 0u, 0u, 0u};
        old1[i] = u32x4_t{0u, 0u, 0u, 0u};
#else
        old0[i] = __builtin_amdgcn_raw_buffer_load_b128(rc, voff[i], 0, 0);
        old1[i] = __builtin_amdgcn_raw_buffer_load_b128(rc, voff[i], 16, 0);
#endif
      }
    }
    if constexpr (COMPUTE) {
#pragma unroll
      for (int b = 0; b < RPP / 32; ++b) {
        const int blk = SPLIT ? b * TM + p : p * (RPP / 32) + b;
        if (wm == blk / TM) {
          const int i = blk % TM;
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
    __builtin_amdgcn_s_barrier();                    // E1: the pass is staged
    if constexpr (!COMPUTE) {
      if (p == 0 && want_cs && tid - 256 < G::BM) {      // waves 4-6: one lane per column of the tile
        const int c = tid - 256, gm = it.m0 + c;
        const float* sp = reinterpret_cast<const float*>(smem + SPARE_OFF);
        const float sum = sp[c];                         // parked by the compute waves of the tile (wn == 0)
        if (gm < P.M) {
          if (plain) P.colsum[gm] += sum; else atomicAdd(P.colsum + gm, sum);
        }
      }
    }
    if (plain) {
      // two iterations at a time (the staged values of all four would not fit beside 144 accumulators)
#pragma unroll
      for (int h = 0; h < ITERS; h += 2) {
        f32x4_t v0[2], v1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = min(r0 + (h + i) * RPI, RPP - 1), x = row & 7;
          v0[i] = *reinterpret_cast<const f32x4_t*>(st + row * G::ROWB + (((2 * c8) ^ x) << 4));
          v1[i] = *reinterpret_cast<const f32x4_t*>(st + row * G::ROWB + (((2 * c8 + 1) ^ x) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v0[i][k] += __uint_as_float(old0[h + i][k]);
            v1[i][k] += __uint_as_float(old1[h + i][k]);
          }
#ifdef HERO_WSB_EPI_NOMEM
        asm volatile("" ::"v"(v0[0]), "v"(v1[0]), "v"(v0[1]), "v"(v1[1]));
        continue;
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v0[i][0]), __float_as_uint(v0[i][1]), __float_as_uint(v0[i][2]), __float_as_uint(v0[i][3])}, rc, voff[h + i], 0, HERO_WS_STORE_DW);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v1[i][0]), __float_as_uint(v1[i][1]), __float_as_uint(v1[i][2]), __float_as_uint(v1[i][3])}, rc, voff[h + i], 16, HERO_WS_STORE_DW);
        }
      }
    } else {
      // k-slice of a tail tile: fp32 atomics, a lane per column (one instruction = 256 contiguous bytes of a row), two
      // rows of the pass per iteration on waves 0-5
      const int col = tid % BN, rr = tid / BN;       // rr 0..1 active (tid < 2 BN)
      const int gc = it.n0 + col;
      if (tid < 2 * BN && gc < P.N) {
        const unsigned sw = (unsigned)(col >> 2);
#pragma unroll 4
        for (int q = 0; q < RPP / 2; ++q) {
          const int row = 2 * q + rr;
          const int gm = it.m0 + tile_row(p, row);
          const float v = *reinterpret_cast<const float*>(st + row * G::ROWB + (((sw ^ (unsigned)(row & 7))) << 4) + (col & 3) * 4);
          if (gm < P.M) atomicAdd(P.C + (size_t)gm * P.ldc + gc, v);
        }
      }
      if (p == G::PASSES - 1) wait_vm<0>();          // this wave's atomics have been acknowledged
    }
    wait_lds();
    __builtin_amdgcn_s_barrier();                    // E2: the slot may be restaged / refilled
  }
  if (!plain && tid == 0)                            // every wave's atomics drained before E2: hand over / reset
    __hip_atomic_store(flags + it.flag, it.order + 1 < it.nslices ? it.order + 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wsb_kernel(WsbArgs g) {
  typedef Geo<3, 3> G;
  constexpr int TM = 3, TN = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;
  int wg;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // the next busy round of this workgroup at or after r (rounds with an idle slot are skipped by all eight waves alike)
  auto next_round = [&](int r) {
    while (r < g.rounds && g.items[(size_t)r * nwg + wg].nk == 0) ++r;
    return r;
  };

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int w = wave - 4;
    constexpr int CA = G::BM / 8, CB = G::BN / 8;
    int lr = next_round(0), ik = 0, lnk = 0;       // round / stage being issued next
    unsigned goa[G::PA], gob[G::PB];
    const char* pa = nullptr;
    const char* pb = nullptr;
    unsigned long long ra_left = 0, rb_left = 0;   // bytes to the end of the operand (64-bit: 1.4 M rows x 3072 columns at config 5)
    unsigned sa = 0, sb = 0, fill = 0;
    auto setup = [&]() {
      const WsbItem it = g.items[(size_t)lr * nwg + wg];
      const WsbProb& P = g.p[it.prob];
      lnk = it.nk;
#pragma unroll
      for (int i = 0; i < G::PA; ++i) {
        const int id = (w * G::PA + i) * 64 + lane, row = id / CA, c = (id % CA) ^ swz_o<G::BM * 2>(row);
        goa[i] = (unsigned)row * (unsigned)P.lda * 2u + (c << 4);
      }
#pragma unroll
      for (int i = 0; i < G::PB; ++i) {
        const int id = (w * G::PB + i) * 64 + lane, row = id / CB, c = (id % CB) ^ swz_o<G::BN * 2>(row);
        gob[i] = (unsigned)row * (unsigned)P.ldb * 2u + (c << 4);
      }
      const int kb = it.k0 * 64;
      pa = reinterpret_cast<const char*>(P.A + (size_t)kb * P.lda + it.m0);
      pb = reinterpret_cast<const char*>(P.B + (size_t)kb * P.ldb + it.n0);
      ra_left = ((unsigned long long)(g.K - kb) * P.lda - it.m0) * 2;
      rb_left = ((unsigned long long)(g.K - kb) * P.ldb - it.n0) * 2;
      sa = 64u * (unsigned)P.lda * 2u;
      sb = 64u * (unsigned)P.ldb * 2u;
    };
    if (lr < g.rounds) setup();
    auto issue = [&]() -> bool {
      if (lr >= g.rounds) return false;
      char* buf = smem + fill;
      // the descriptor base moves with the k-step, so offsets stay small; only the range is clamped to 32 bits
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pa), 0, (unsigned)(ra_left < 0xfffffff0ull ? ra_left : 0xfffffff0ull), 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, (unsigned)(rb_left < 0xfffffff0ull ? rb_left : 0xfffffff0ull), 0x00020000);
#ifndef HERO_WSB_NOLOADS        // lab ablations (tools/lab/build_variants.sh): results are garbage, timing only
#pragma unroll
      for (int i = 0; i < G::PA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, HERO_LDS_PTR(buf + (w * G::PA + i) * 1024), 16, goa[i], 0, 0, HERO_WS_LOAD_AUX_A);
#pragma unroll
      for (int i = 0; i < G::PB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, HERO_LDS_PTR(buf + G::A_BYTES + (w * G::PB + i) * 1024), 16, gob[i], 0, 0, HERO_WS_LOAD_AUX_B);
#else
      (void)ra; (void)rb; (void)buf;
#endif
      fill += G::STAGE;
      if (fill == NS * G::STAGE) fill = 0;
      if (++ik == lnk) {
        ik = 0;
        lr = next_round(lr + 1);
        if (lr < g.rounds) setup();
      } else {
        pa += sa; pb += sb;
        ra_left = ra_left > sa ? ra_left - sa : 0ull;
        rb_left = rb_left > sb ? rb_left - sb : 0ull;
      }
      return true;
    };
    issue();
    const bool second = issue();
    if (second) wait_vm<G::PW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                                     // B(-1): stage 0 landed
    unsigned slot = 0;
    // Bias gradients (WsbProb.colsum): the column sums of the dY panel come from the COMPUTE waves (one extra MFMA per row
    // block against a constant selector, see below); the loader waves only apply them in the epilogue.  Rounds 4-5 had
    // these waves add the landed stages up between their DMA issues - 24 KB more LDS reads per k-step on a loop that is
    // bound by the LDS: the tiles that did it ran ~20 % slower and gated their round (profiles/r06_d4_ride_ab.txt).
    for (int r = next_round(0); r < g.rounds; r = next_round(r + 1)) {
      const WsbItem it = g.items[(size_t)r * nwg + wg];
      const bool want_cs = it.n0 == 0 && g.p[it.prob].colsum != nullptr;         // uniform
      for (int t = 0; t < it.nk; ++t) {
        const bool more = issue();
        if (more) wait_vm<G::PW>(); else wait_vm<0>();
#ifndef HERO_WSB_NOBAR
        __builtin_amdgcn_s_barrier();                                 // B(u)
#endif
        if (t + 1 < it.nk) { slot += G::STAGE; if (slot == NS * G::STAGE) slot = 0; }
      }
#ifndef HERO_WSB_NOEPI
      epilogue_acc<G, false>(g.p[it.prob], it, g.flags, smem, slot, nullptr, wave, lane, want_cs);
#endif
      slot += G::STAGE; if (slot == NS * G::STAGE) slot = 0;
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int wm = wave >> 1, wn = wave & 1;
  const int arow0 = wm * TM * 32, brow0 = wn * TN * 32;
  unsigned ao[TM], bo[TN];
  {
    const int p = lane & 15, gq = (lane >> 4) & 1, kg = lane >> 5;
    const int krow = kg * 8 + (p >> 2);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int col = arow0 + i * 32 + gq * 16 + 4 * (p & 3);
      ao[i] = krow * (G::BM * 2) + ((((col >> 3) ^ swz_o<G::BM * 2>(krow)) << 4) | ((col & 7) * 2));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = brow0 + j * 32 + gq * 16 + 4 * (p & 3);
      bo[j] = G::A_BYTES + krow * (G::BN * 2) + ((((col >> 3) ^ swz_o<G::BN * 2>(krow)) << 4) | ((col & 7) * 2));
    }
  }
  typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
  bf16x8_t a0[TM], b0[TN], a1[TM], b1[TN];
  auto ldf = [&](bf16x8_t (&a)[TM], bf16x8_t (&b)[TN], const char* st, int ks) {       // order a[0], b[..], a[1..]: see gemm_ws_kernel
#ifdef HERO_WSB_NOLDF
    return;
#endif
    auto ra = [&](int i) {
      const char* q = st + ao[i] + ks * 16 * (G::BM * 2);
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (G::BM * 2)));
      a[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    ra(0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const char* q = st + bo[j] + ks * 16 * (G::BN * 2);
      const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q));
      const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(q + 4 * (G::BN * 2)));
      b[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 1; i < TM; ++i) ra(i);
  };
  f32x16_t acc[TM][TN];
  // Bias gradient = column sums of the dY panel (the B operand a[i]: lane <-> dW row): one more MFMA per row block against
  // a constant SELECTOR as the A operand - sel_i[mm][k] = 1 for the eight output rows mm = 8 i .. 8 i + 7, else 0 - so that
  // rows 8 i .. 8 i + 7 of ONE extra accumulator collect block i's sums: accumulator register 4 i of lane l < 32 = the sum
  // of column arow0 + 32 i + l of the tile.  Only the waves with wn == 0 of the tiles with n0 == 0: 3 MFMAs on top of 9 per
  // 16-k slice on a loop whose matrix pipe is about half idle, no LDS traffic, 16 more registers.
  f32x16_t accb;
  unsigned selw[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) selw[i] = ((lane & 31) >> 3) == i ? 0x3f803f80u : 0u;       // two bf16 ones
  auto mma = [&](const bf16x8_t (&a)[TM], const bf16x8_t (&b)[TN], auto cs) {
#ifdef HERO_WSB_NOMFMA
    asm volatile("" ::"v"(a[0]), "v"(b[0]), "v"(a[TM - 1]), "v"(b[TN - 1]));
    return;
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // dW^T: lane <-> output row
      if constexpr (decltype(cs)::value) {
        const u32x4_t w4 = {selw[i], selw[i], selw[i], selw[i]};
        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w4), a[i], accb, 0, 0, 0);
      }
    }
  };
  // the k-loop of one item (two instantiations: a branch inside the loop would split its scheduling regions)
  unsigned curo = 0, last = 0;
  auto k_loop = [&](int nk, auto cs) {
    constexpr int NM = TM * TN + (decltype(cs)::value ? TM : 0);
    for (int t = 0; t < nk; ++t) {
      const char* cur = smem + curo;
      last = curo;
      curo += G::STAGE;
      if (curo == NS * G::STAGE) curo = 0;
      const char* nxt = smem + curo;
      // One scheduling region per 16-k slice: the 12 transposing fragment reads of the NEXT slice are spread between
      // the 9 MFMAs of the current one (MFMA, 2 reads, MFMA, 2 reads, ...).  Issued as a block in front of the MFMAs
      // (round 2) the reads cost ~140 cycles of MFMA-idle issue time per slice: 0.89 us per 64-k step with the DMA
      // switched off against 0.58 us of MFMA issue (tools/lab/wsb_sweep.py, noloads).
      ldf(a1, b1, cur, 1);
      mma(a0, b0, cs);
      WS_INTERLEAVE(NM, 2 * (TM + TN));
      __builtin_amdgcn_sched_barrier(0);
      ldf(a0, b0, cur, 2);
      mma(a1, b1, cs);
      WS_INTERLEAVE(NM, 2 * (TM + TN));
      __builtin_amdgcn_sched_barrier(0);
      ldf(a1, b1, cur, 3);
      mma(a0, b0, cs);
      WS_INTERLEAVE(NM, 2 * (TM + TN));
      __builtin_amdgcn_sched_barrier(0);
      wait_lds();
#ifndef HERO_WSB_NOBAR
      __builtin_amdgcn_s_barrier();                                   // B(u)
#endif
      __builtin_amdgcn_sched_barrier(0);
      ldf(a0, b0, nxt, 0);            // unconditional (a branch around it doubles the MFMA code and spills): after the
      mma(a1, b1, cs);                // item's last step this reads the landed first stage of the next item and is
      WS_INTERLEAVE(NM, 2 * (TM + TN));        // simply read again behind the epilogue
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  __builtin_amdgcn_s_setprio(2);
  __builtin_amdgcn_s_barrier();                                       // B(-1)
  int r = next_round(0);
  if (r < g.rounds) ldf(a0, b0, smem, 0);
  while (r < g.rounds) {
    const WsbItem it = g.items[(size_t)r * nwg + wg];
    const int rn = next_round(r + 1);
    const bool cs_on = wn == 0 && it.n0 == 0 && g.p[it.prob].colsum != nullptr;        // uniform
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    last = curo;
    if (cs_on) {
#pragma unroll
      for (int e = 0; e < 16; ++e) accb[e] = 0.f;
      k_loop(it.nk, std::true_type{});
      // park the sums for the epilogue (loader waves 0-2 apply them behind the first pass barrier)
      float* sp = reinterpret_cast<float*>(smem + SPARE_OFF);
      if (lane < 32) {
#pragma unroll
        for (int i = 0; i < TM; ++i) sp[arow0 + 32 * i + lane] = accb[4 * i];
      }
    } else {
      k_loop(it.nk, std::false_type{});
    }
    __builtin_amdgcn_s_setprio(0);
#ifndef HERO_WSB_NOEPI
    epilogue_acc<G, true>(g.p[it.prob], it, g.flags, smem, last, acc, wave, lane);
#else
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
#endif
    __builtin_amdgcn_s_setprio(2);
    if (rn < g.rounds) ldf(a0, b0, smem + curo, 0);
    r = rn;
  }
}

// explicit instantiations (hipcc does not emit the host stubs of kernels that are only reached through two
// levels of host-side templates)
#define HERO_WS_INST(TM, TN)                                                                  \
  template __global__ void gemm_ws_kernel<TM, TN, false, 0>(WsArgs);                          \
  template __global__ void gemm_ws_kernel<TM, TN, false, EK_BIAS>(WsArgs);                    \
  template __global__ void gemm_ws_kernel<TM, TN, false, EK_BIAS | EK_RES | EK_DROP>(WsArgs); \
  template __global__ void gemm_ws_kernel<TM, TN, false, EK_BIAS | EK_GELU>(WsArgs);          \
  template __global__ void gemm_ws_kernel<TM, TN, false, EK_RES>(WsArgs);                     \
  template __global__ void gemm_ws_kernel<TM, TN, false, EK_GELU_BWD>(WsArgs);                \
  template __global__ void gemm_ws_kernel<TM, TN, true, 0>(WsArgs);
HERO_WS_INST(3, 3)
HERO_WS_INST(2, 3)

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
__global__ void ws_scale_f32_kernel(float* c, int M, int N, int ldc, float beta) {
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

static int num_cus() {
  static int n = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : 256;
  }();
  return n;
}

template <int TM, int TN, bool TR, int EK>
static int launch(WsArgs g, int slot, hipStream_t s) {
  typedef Geo<TM, TN> G;
  HERO_ENSURE_LDS((&gemm_ws_kernel<TM, TN, TR, EK>), G::LDS, "gemm_ws_kernel");
  const int grid = g.nwork < num_cus() ? g.nwork : num_cus();
  void* tok = gemm_prof_begin(slot, s);
  hipLaunchKernelGGL((gemm_ws_kernel<TM, TN, TR, EK>), dim3(grid), dim3(512), G::LDS, s, g);
  gemm_prof_end(tok, 2.0 * (double)g.M * (double)g.N * (double)g.K, s);
  return check_launch("hero_gemm(ws)");
}

template <int TM, int TN>
static int launch_kk(const WsArgs& g, hipStream_t s) {
  const HeroGemmEpilogue& e = g.epi;
  const bool b = e.bias != nullptr, r = e.residual != nullptr, d = e.dropout.threshold16 != 0;
  if (e.colsum != nullptr && e.act != HERO_ACT_GELU_BWD && e.act != HERO_ACT_MUL_AUX) return -1;     // column sums exist in the gelu' epilogue only: 4-wave path
  if (e.act == HERO_ACT_NONE && b && !r && !d) return launch<TM, TN, false, EK_BIAS>(g, TM == 3 ? 8 : 10, s);
  if (e.act == HERO_ACT_NONE && b && r) return launch<TM, TN, false, EK_BIAS | EK_RES | EK_DROP>(g, TM == 3 ? 8 : 10, s);
  if ((e.act == HERO_ACT_GELU || e.act == HERO_ACT_GELU_DG) && b && !r && !d) return launch<TM, TN, false, EK_BIAS | EK_GELU>(g, TM == 3 ? 8 : 10, s);
  if (e.act == HERO_ACT_RELU && b && !r && !d && e.aux) return launch<TM, TN, false, EK_BIAS | EK_GELU>(g, TM == 3 ? 8 : 10, s);
  if constexpr (TM == 1) {         // LinearLayer (frame_transform, model/layers.py:86-93 + model/model.py:211-212): relu(x W^T + b) + matched features
    if (e.act == HERO_ACT_RELU && b && r && !d && e.aux) return launch<TM, TN, false, EK_BIAS | EK_GELU | EK_RES>(g, 10, s);
  }
  if (e.act == HERO_ACT_NONE && !b && !r && !d) return launch<TM, TN, false, 0>(g, TM == 3 ? 8 : 10, s);
  if (e.act == HERO_ACT_NONE && !b && r && !d) return launch<TM, TN, false, EK_RES>(g, TM == 3 ? 8 : 10, s);
  if ((e.act == HERO_ACT_GELU_BWD || e.act == HERO_ACT_MUL_AUX) && !b && !r && !d) return launch<TM, TN, false, EK_GELU_BWD>(g, TM == 3 ? 8 : 10, s);
  return -1;
}

}  // namespace ws

// Problems this family takes: bf16, K,K operands with one of the six hot-path epilogues, or the O,O
// wgrad accumulate; large enough to fill the chip with 192 x 192 (or, under one round, 128 x 192) tiles.
// force_cfg: -1 heuristic, 8 never, 9 always 192 x 192, 10 always 128 x 192, 13 / 14 always 64 x 128 / 64 x 192 (when the shape
// is legal).  (Round 4's deferred-epilogue variant - the finished tile parked in the loader waves' registers and drained
// during the next main loop, bit-exact and 1.0-1.3x SLOWER, profiles/r04_wsd_ab.txt - lives in tools/lab/gemm_wsd.hip.)
int gemm_ws_run(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int a_layout,
                int b_layout, const HeroGemmEpilogue& epi, int force_cfg, hipStream_t s) {
  using namespace ws;
  if (force_cfg == 8) return -1;
  if ((force_cfg < 9 || force_cfg > 14 || force_cfg == 11 || force_cfg == 12) && force_cfg != -1) return -1;      // a forced 4-wave geometry
  typedef Geo<3, 3> G;
  const bool kk = a_layout == HERO_LAYOUT_K && b_layout == HERO_LAYOUT_K;
  const bool oo = a_layout == HERO_LAYOUT_O && b_layout == HERO_LAYOUT_O;
  if (!kk && !oo) return -1;
  if (K < 64 || N % 8 != 0 || M < 8) return -1;
  if ((force_cfg == 10 || force_cfg == 13 || force_cfg == 14) && !kk) return -1;
  WsArgs g;
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tiles_m = (M + G::BM - 1) / G::BM;
  g.tiles_n = (N + G::BN - 1) / G::BN;
  g.epi = epi;
  const int ntile = g.tiles_m * g.tiles_n;
  const int cus = num_cus();
  if (kk) {
    if (K % 64 != 0 || epi.out_f32 || epi.split_k > 1) return -1;
    // every per-lane offset of this family is relative to its tile (operand panels, residual / saved pre-activation reads,
    // stores); only the weight matrix B is addressed from its base
    if ((size_t)N * ldb * 2 >= 0x7fffffffull || (size_t)192 * lda * 2 >= 0x7fffffffull || (size_t)192 * ldc * 2 >= 0x7fffffffull) return -1;
    g.nsplit = 1;
    g.k_per_split = K;
    g.group = 8;
    // Geometry by rounds: the persistent workgroups walk ceil(tiles / CUs) tiles each, and a 128 x 192 tile (Geo<2, 3>)
    // costs ~0.75-0.8 of a 192 x 192 one (tools/lab/geo_ab.py: 11 vs 15 us per round at K = 768, 30 vs 40 at K = 3072).
    // The smaller tile wins where it does not need proportionally more rounds:
    //  * under one round (M = 1920, N = 3072: 240 instead of 160 tiles; N = 2304: 180 instead of 120);
    //  * just over a round boundary of the 192 x 192 grid - the usual case of a ragged batch: M = 14450 rows, N = 768
    //    = 304 tiles = 2 rounds at 59 % against 452 tiles of 128 x 192 = 2 rounds at 88 %: 32.4 -> 25.4 us (K = 768),
    //    80.7 -> 67.0 (K = 3072).  The bench batch sits ON the boundary (12000 rows x 768 = 252 tiles) and keeps 192 x 192.
    typedef Geo<2, 3> G2;
    const int t23 = ((M + G2::BM - 1) / G2::BM) * g.tiles_n;
    const int r33 = (ntile + cus - 1) / cus, r23 = (t23 + cus - 1) / cus;
    const bool small = force_cfg == 10 || (force_cfg == -1 && r23 * 4 < r33 * 5 && t23 * 5 >= cus * 2);
    if (small) {
      g.tiles_m = (M + G2::BM - 1) / G2::BM;
      g.nwork = t23;
      return launch_kk<2, 3>(g, s);
    }
    // Under half a round of 192 x 192 tiles - the Temporal Transformer's 1920 rows into N = 768: 40 tiles - the 64-row
    // geometries: 64 x 128 tiles (180 workgroups, 24 KB per step, a 6-deep ring so that as many bytes are in flight as the
    // large tiles keep with three stages).  The 4-wave 64 x 64 kernels put 360 workgroups of 16 KB steps on 256 CUs and are
    // bound by the busiest CU's L2 -> LDS fill (two tiles x 48 steps: 26 us at K = 3072, the vendor library 17.7,
    // profiles/r04_vs_library.txt); 18.2 us with this geometry, 6.7 vs 8.0 at K = 768 (tools/lab/smallm_ws.py).
    if (force_cfg == 13 || force_cfg == 14 || (force_cfg == -1 && ntile * 2 < cus && K >= 512)) {
      typedef Geo<1, 2> G12;
      typedef Geo<1, 3> G13;
      const int rows64 = (M + 63) / 64;
      const int t12 = rows64 * ((N + G12::BN - 1) / G12::BN), t13 = rows64 * g.tiles_n;
      // 64 x 192 (four stages) where 64 x 128 tiles would spill into a second round: the ragged batch's ~3100 padded frame rows
      // (49 x 6 = 294 tiles against 49 x 4 = 196); per tile it is ~20 % behind 64 x 128, a partial second round costs more
      if (force_cfg == 14 || (force_cfg == -1 && t12 > cus && t13 <= cus)) {
        g.tiles_m = rows64; g.nwork = t13;
        return launch_kk<1, 3>(g, s);
      }
      if (force_cfg == 13 || (t12 * 8 >= cus && t12 <= cus)) {        // down to the 480 query rows (48 tiles: 12.0 vs 13.7 us at K = 2304)
        g.tiles_m = rows64; g.tiles_n = (N + G12::BN - 1) / G12::BN; g.nwork = t12;
        return launch_kk<1, 2>(g, s);
      }
    }
    // worth it from about half a round of tiles; below that the 4-wave 64 x 64 / 128 x 128 tiles fill the chip better
    if (force_cfg != 9 && ntile * 2 < cus) return -1;
    g.nwork = ntile;
    return launch_kk<3, 3>(g, s);
  }
  // O,O: fp32 accumulate, reduction split so that the items fill the CUs
  if (!epi.out_f32 || epi.act != HERO_ACT_NONE || epi.bias || epi.residual || epi.dropout.threshold16 != 0 || epi.colsum) return -1;
  if (M % 8 != 0) return -1;
  if ((size_t)K * lda * 2 >= 0xffffffffull || (size_t)K * ldb * 2 >= 0xffffffffull) return -1;
  const int ksteps = (K + 63) / 64;
  // The fp32-atomic merge costs ~split x output bytes at ~1.2 TB/s (measured: 31 us for 4 x 9.4 MB) and is paid
  // after the last k-step by every item at once: worth it only where few splits fill the chip (>= 64 tiles of
  // 192 x 192, i.e. the 3072 x 768 weights: 77 vs 84 us); smaller outputs stay on the 128 x 128 kernel's finer
  // split.  (Tried and dropped: 96 x 96 tiles with the reduction split across the four compute waves of a
  // workgroup and no cross-workgroup merge - 104 vs 82 us: four times the LDS-DMA bytes per flop, and the
  // direct-to-LDS path of a CU saturates near 50 GB/s while MFMAs and fragment reads are running.)
  if (force_cfg != 9 && (ksteps < 32 || ntile < 64)) return -1;
  int split = cus / ntile;                               // at most one round of items
  if (split > ksteps / 4) split = ksteps / 4;
  if (split < 1) split = 1;
  const int per = (ksteps + split - 1) / split;
  split = (ksteps + per - 1) / per;
  g.nsplit = split;
  g.k_per_split = per * 64;
  g.nwork = ntile * split;
  g.group = 8;
  if (epi.beta != 1.f) {
    hipLaunchKernelGGL(ws_scale_f32_kernel, dim3(1024), dim3(256), 0, s, static_cast<float*>(C), M, N, ldc, epi.beta);
    const int rc = check_launch("hero_gemm(ws scale)");
    if (rc) return rc;
  }
  return launch<3, 3, true, 0>(g, 9, s);
}

}  // namespace hero

#ifdef HERO_WS_TRACE
extern "C" int hero_ws_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hero::ws::g_ws_trace), sizeof(unsigned long long) * 4 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif

// dW_p += dY_p^T X_p for up to 4 problems over the same K rows, one stream-K launch (hero_hip.h).
extern "C" int hero_wgrad_group(const HeroWgradProblem* probs, int n, int K, int dtype, hero_stream_t stream) {
  using namespace hero;
  using namespace hero::ws;
  HERO_REQUIRE(probs && n >= 1 && n <= 4, "hero_wgrad_group: 1..4 problems");
  HERO_REQUIRE(dtype == HERO_BF16 || dtype == HERO_F32, "hero_wgrad_group: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  typedef Geo<3, 3> G;
  bool ok = dtype == HERO_BF16 && K >= 64 * 16 && gemm_forced_config() != 8;
  WsgArgs g;
  int tiles = 0;
  for (int i = 0; i < n && ok; ++i) {
    const HeroWgradProblem& q = probs[i];
    HERO_REQUIRE(q.dy && q.x && q.dw && q.M > 0 && q.N > 0, "hero_wgrad_group: bad problem %d", i);
    HERO_REQUIRE(q.dbias == nullptr, "hero_wgrad_group: dbias is a hero_wgrad_batch feature (problem %d)", i);
    ok = ok && q.M % 8 == 0 && q.N % 8 == 0 && q.ld_dy % 8 == 0 && q.ld_x % 8 == 0 && q.ld_dw % 4 == 0 &&
         (((uintptr_t)q.dy | (uintptr_t)q.x | (uintptr_t)q.dw) & 15) == 0 &&
         (size_t)K * q.ld_dy * 2 < 0xffffffffull && (size_t)K * q.ld_x * 2 < 0xffffffffull;
    WsgProb& P = g.p[i];
    P.A = static_cast<const bf16_t*>(q.dy); P.B = static_cast<const bf16_t*>(q.x); P.C = q.dw;
    P.M = q.M; P.N = q.N; P.lda = q.ld_dy; P.ldb = q.ld_x; P.ldc = q.ld_dw;
    P.tiles_n = (q.N + G::BN - 1) / G::BN;
    P.tile0 = tiles;
    tiles += ((q.M + G::BM - 1) / G::BM) * P.tiles_n;
  }
  const int ksteps = (K + 63) / 64;
  const int cus = num_cus();
  if (ok && (long long)tiles * ksteps >= 8LL * cus) {
    for (int i = n; i < 4; ++i) g.p[i] = g.p[0];
    g.nprob = n; g.K = K; g.ksteps = ksteps; g.total_tiles = tiles;
    HERO_ENSURE_LDS(&gemm_wsg_kernel, G::LDS, "gemm_wsg_kernel");
    double flops = 0.0;
    for (int i = 0; i < n; ++i) flops += 2.0 * probs[i].M * (double)probs[i].N * K;
    void* tok = gemm_prof_begin(9, s);
    hipLaunchKernelGGL(gemm_wsg_kernel, dim3(cus), dim3(512), G::LDS, s, g);
    gemm_prof_end(tok, flops, s);
    return check_launch("hero_wgrad_group");
  }
  // outside the kernel's envelope: one hero_gemm per problem (accumulate)
  for (int i = 0; i < n; ++i) {
    const HeroWgradProblem& q = probs[i];
    HeroGemmEpilogue e = {};
    e.out_f32 = 1; e.beta = 1.f; e.split_k = q.split_hint > 0 ? q.split_hint : 1; e.dropout.scale = 1.f;
    const int rc = hero_gemm(q.dy, q.x, q.dw, q.M, q.N, K, q.ld_dy, q.ld_x, q.ld_dw, HERO_LAYOUT_O, HERO_LAYOUT_O, dtype, &e, stream);
    if (rc) return rc;
  }
  return HERO_OK;
}

// ------------------------------------------------------------------------------------------------
// hero_wgrad_batch: plan + launch (hero_hip.h)
// ------------------------------------------------------------------------------------------------
namespace hero {
namespace ws {
__device__ int g_wsb_flags[512];          // slice-order flags of the tail tiles; every launch leaves them zero
constexpr int PLAN_MAGIC = 0x57534232;    // "WSB2"
constexpr int PLAN_HDR = 8;               // words: magic, nwg, rounds, items, n problems, K, tiles, tail slices
}  // namespace ws
}  // namespace hero

extern "C" int hero_wgrad_batch_plan(const HeroWgradProblem* probs, int n, int K, int32_t* plan, int capacity_words) {
  using namespace hero;
  using namespace hero::ws;
  typedef Geo<3, 3> G;
  HERO_REQUIRE(probs && n >= 1 && n <= HERO_WGRAD_BATCH_MAX, "hero_wgrad_batch_plan: 1..%d problems", HERO_WGRAD_BATCH_MAX);
  HERO_REQUIRE(K >= 1, "hero_wgrad_batch_plan: K = %d", K);
  const int nwg = num_cus(), ksteps = (K + 63) / 64;
  if (nwg % 8 != 0 || ksteps < 8) return 0;            // short reductions: the per-group path (hero_wgrad_group / hero_gemm)
  // tiles in patch order: up to 8 x 4 (or tiles_m x 32 / tiles_m) tiles of one problem are consecutive = one XCD's round
  struct T3 { int prob, mt, nt; };
  std::vector<T3> tiles;
  for (int i = 0; i < n; ++i) {
    HERO_REQUIRE(probs[i].M > 0 && probs[i].N > 0 && probs[i].M % 8 == 0 && probs[i].N % 8 == 0, "hero_wgrad_batch_plan: bad problem %d", i);
    const int tm = (probs[i].M + G::BM - 1) / G::BM, tn = (probs[i].N + G::BN - 1) / G::BN;
    const int per = nwg / 8;
    int pm = tm < 8 ? tm : 8;
    int pn = per / pm > 0 ? per / pm : 1;
    if (pn > tn) { pn = tn; pm = per / pn < tm ? per / pn : tm; if (pm < 1) pm = 1; }
    for (int bm = 0; bm < tm; bm += pm)
      for (int bn = 0; bn < tn; bn += pn)
        for (int mi = bm; mi < bm + pm && mi < tm; ++mi)
          for (int ni = bn; ni < bn + pn && ni < tn; ++ni) tiles.push_back({i, mi, ni});
  }
  const int T = (int)tiles.size();
  const int full = T / nwg, rem = T % nwg;
  // (round 3 left groups of fewer than nwg / 4 tiles to the stream-K group kernel, whose fp32 atomics land in any order;
  // the sliced tail keeps them ordered, and such groups are a few microseconds either way)
  // Tail (tiles % nwg != 0): the remaining tiles are dealt to the XCDs (per_xcd each) and every XCD balances its p tiles
  // over its c = nwg / 8 workgroups.  Each workgroup gets a quota of q = ceil(p ksteps / c) k-steps:
  //   * every tile is cut into S = floor(c / p) "big" slices of q steps on S x p workgroups (round 0 of the tail);
  //   * what is left of each tile (r = ksteps - S q steps) is packed, tile after tile, into the L = c - S p workgroups
  //     that are still free, up to q steps each - one piece per extra tail round, a piece may be cut at a workgroup boundary.
  // Round 3 had the equal slices only (S = floor(c / p), L workgroups idle): 192 tiles - ONE BertLayer, which is all the
  // byte cap of the queue lets config 5 batch - ran on 192 of 256 workgroups.  Now every workgroup carries ~q steps.
  // Slice order (the fp32 atomics of a tile are applied in this order, which keeps dW bit-reproducible) follows the
  // time a piece completes when nobody waits: the head piece of a cut remainder first, then its other piece, then the
  // big slices - so a waiting piece only ever waits for pieces that finish earlier, and the k-ranges are laid out in
  // the same order (remainder pieces at the start of the reduction, big slices behind them).
  struct Piece { int slot, round, tile, k0, nk, order, nslices; };
  std::vector<Piece> pieces;
  int tail_rounds = rem ? 1 : 0, S = 1;
  const int per_xcd_cap = nwg / 8, per_xcd = rem ? (rem + 7) / 8 : 0;
  if (rem) {
    HERO_REQUIRE(per_xcd <= 512 / 8, "hero_wgrad_batch_plan: flag capacity");
    for (int xcd = 0; xcd * per_xcd < rem; ++xcd) {
      const int t0 = xcd * per_xcd, p = (rem - t0 < per_xcd) ? rem - t0 : per_xcd, c = per_xcd_cap;
      // equal slices first (round 3's rule: at least 4 k-steps per slice, at most 8 slices - every slice adds a tile of atomics)
      int Sx = c / p;
      while (Sx > 1 && ksteps / Sx < 4) --Sx;
      if (Sx > 8) Sx = 8;
      // Round 6 (profiles/r06_wgrad_ceiling.txt: the 92 tiles of the 4352-wide projection gradient, 30 k-steps, ran 59 us in two
      // slices where their main loops take 18): a sliced tile pays ~16 us of ordered fp32 atomics PER SLICE - slice s waits for
      // slice s - 1 to drain - where a whole tile pays ~8 us of plain read-add-write, and a k-step is ~1.2 us.  Slices only
      // where they save more k-steps than that: short reductions (the 1920-row and 480-row groups) keep whole tiles.
      auto cost_us = [&](int sl) { return (double)((ksteps + sl - 1) / sl) * 1.17 + (sl == 1 ? 8.0 : 9.0 + 16.0 * sl); };
      while (Sx > 1 && cost_us(Sx - 1) <= cost_us(Sx)) --Sx;
      const int L = c - Sx * p;
      // workgroups left over (c / p not whole) take the remainders - where the reduction is long enough that a piece's own
      // epilogue (a tile of atomics, ~4 k-steps' worth) does not eat the gain: ksteps >= 128, pieces of >= 4 steps
      // (a piece costs its k-steps + E for its own epilogue, so the quota q solves p (ksteps - Sx q + E) = L (q + E); worth
      // it only where the remainders are a real share of a tile - 30 tail tiles on 32 workgroups stay as they are)
      const int E = 6;
      int q = (int)(((long long)p * ksteps + (long long)(p - L) * E + c - 1) / c);
      int r = ksteps - Sx * q;
      if (L == 0 || Sx != c / p || ksteps < 128 || r * 8 < ksteps) { r = 0; q = ksteps; }
      if (xcd == 0) S = Sx;
      // remainder pieces: walk the free workgroups
      std::vector<std::vector<Piece>> of_tile(p);
      int slot = Sx * p, fillq = 0, rnd = 0;
      for (int t = 0; t < p && r > 0; ++t) {
        int left = r;
        std::vector<Piece> mine;
        while (left > 0) {
          int room = q - fillq;
          if (slot >= c - 1) room = left;             // the last free workgroup takes what is left (rounding)
          else if (room < 2 || (room < left && left - room < 2)) {
            if (room < 2) { ++slot; fillq = 0; rnd = 0; continue; }
            room = left;                               // do not leave a 1-step crumb for the next workgroup
          }
          const int take = left < room ? left : room;
          mine.push_back({slot < c ? slot : c - 1, rnd, t, 0, take, 0, 0});
          left -= take; fillq += take; ++rnd;
          if (fillq >= q && slot < c - 1) { ++slot; fillq = 0; rnd = 0; }
        }
        // completion order: a later-emitted piece sits at the head of the next workgroup and finishes first
        int k0 = 0, ord = 0;
        for (int i = (int)mine.size() - 1; i >= 0; --i) { mine[i].k0 = k0; mine[i].order = ord++; k0 += mine[i].nk; }
        of_tile[t] = mine;
      }
      for (int t = 0; t < p; ++t) {
        const int nrem = (int)of_tile[t].size();
        const int big0 = r;                            // big slices cover [r, ksteps)
        for (int sl = 0; sl < Sx; ++sl) {
          const int k0 = big0 + (int)((long long)(ksteps - big0) * sl / Sx), k1 = big0 + (int)((long long)(ksteps - big0) * (sl + 1) / Sx);
          of_tile[t].push_back({sl * p + t, 0, t, k0, k1 - k0, nrem + sl, 0});
        }
        for (Piece& pc : of_tile[t]) {
          pc.nslices = (int)of_tile[t].size();
          pc.slot += xcd * per_xcd_cap;
          pc.tile = t0 + t;
          if (pc.round + 1 > tail_rounds) tail_rounds = pc.round + 1;
          pieces.push_back(pc);
        }
      }
    }
  }
  const int rounds = full + tail_rounds;
  const int words = PLAN_HDR + rounds * nwg * 8;
  HERO_REQUIRE(plan && capacity_words >= words, "hero_wgrad_batch_plan: needs %d words, capacity %d", words, capacity_words);
  for (int i = PLAN_HDR; i < words; ++i) plan[i] = 0;
  auto item = [&](int r, int w) { return plan + PLAN_HDR + ((size_t)r * nwg + w) * 8; };
  for (int r = 0; r < full; ++r)
    for (int w = 0; w < nwg; ++w) {
      const T3& t = tiles[(size_t)r * nwg + w];
      int32_t* q = item(r, w);
      q[0] = t.prob; q[1] = t.mt * G::BM; q[2] = t.nt * G::BN; q[3] = 0; q[4] = ksteps; q[5] = 0; q[6] = 1; q[7] = 0;
    }
  for (const Piece& pc : pieces) {
    const T3& t = tiles[(size_t)full * nwg + pc.tile];
    int32_t* q = item(full + pc.round, pc.slot);
    HERO_REQUIRE(q[4] == 0, "hero_wgrad_batch_plan: internal error (slot used twice)");
    q[0] = t.prob; q[1] = t.mt * G::BM; q[2] = t.nt * G::BN; q[3] = pc.k0; q[4] = pc.nk; q[5] = pc.order; q[6] = pc.nslices;
    q[7] = (pc.slot / per_xcd_cap) * (512 / 8) + (pc.tile % per_xcd);
  }
  plan[0] = PLAN_MAGIC; plan[1] = nwg; plan[2] = rounds; plan[3] = rounds * nwg; plan[4] = n; plan[5] = K; plan[6] = T; plan[7] = S;
  return words;
}

extern "C" int hero_wgrad_batch(const HeroWgradProblem* probs, int n, int K, int dtype, const int32_t* plan_dev, int plan_words,
                                hero_stream_t stream) {
  using namespace hero;
  using namespace hero::ws;
  typedef Geo<3, 3> G;
  HERO_REQUIRE(probs && n >= 1 && n <= HERO_WGRAD_BATCH_MAX, "hero_wgrad_batch: 1..%d problems", HERO_WGRAD_BATCH_MAX);
  HERO_REQUIRE(dtype == HERO_BF16, "hero_wgrad_batch: bf16 only (dtype %d)", dtype);
  HERO_REQUIRE(plan_dev && plan_words > PLAN_HDR && (plan_words - PLAN_HDR) % (8 * num_cus()) == 0,
               "hero_wgrad_batch: the plan (%d words) was not made for this device (%d workgroups)", plan_words, num_cus());
  HERO_REQUIRE(((uintptr_t)plan_dev & 15) == 0, "hero_wgrad_batch: plan_dev must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  WsbArgs g;
  double flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const HeroWgradProblem& q = probs[i];
    HERO_REQUIRE(q.dy && q.x && q.dw && q.M > 0 && q.N > 0 && q.M % 8 == 0 && q.N % 8 == 0 && q.ld_dy % 8 == 0 && q.ld_x % 8 == 0 &&
                     q.ld_dw % 4 == 0 && (((uintptr_t)q.dy | (uintptr_t)q.x | (uintptr_t)q.dw) & 15) == 0 &&
                     (size_t)64 * q.ld_dy * 2 < 0x7fffffffull && (size_t)64 * q.ld_x * 2 < 0x7fffffffull &&
                     (size_t)q.M * q.ld_dw * 4 < 0x7ffffff0ull,
                 "hero_wgrad_batch: problem %d is unaligned / too large", i);
    WsbProb& P = g.p[i];
    P.A = static_cast<const bf16_t*>(q.dy); P.B = static_cast<const bf16_t*>(q.x); P.C = q.dw;
    P.M = q.M; P.N = q.N; P.lda = q.ld_dy; P.ldb = q.ld_x; P.ldc = q.ld_dw; P.pad_ = 0;
    P.colsum = q.dbias;
    flops += 2.0 * q.M * (double)q.N * K;
  }
  for (int i = n; i < HERO_WGRAD_BATCH_MAX; ++i) g.p[i] = g.p[0];
  g.items = reinterpret_cast<const WsbItem*>(plan_dev + PLAN_HDR);
  g.rounds = (plan_words - PLAN_HDR) / (8 * num_cus());
  g.K = K; g.nprob = n; g.pad_ = 0;
  // resolved once PER DEVICE (a __device__ symbol has one address per device), outside any stream capture of later calls
  static int* flags_of[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  HERO_REQUIRE(dev >= 0 && dev < 64, "hero_wgrad_batch: device %d", dev);
  if (!flags_of[dev]) HERO_REQUIRE(hipGetSymbolAddress(reinterpret_cast<void**>(&flags_of[dev]), HIP_SYMBOL(hero::ws::g_wsb_flags)) == hipSuccess,
                                   "hero_wgrad_batch: flag storage");
  g.flags = flags_of[dev];
  HERO_ENSURE_LDS(&gemm_wsb_kernel, G::LDS, "gemm_wsb_kernel");
  void* tok = gemm_prof_begin(9, s);
  hipLaunchKernelGGL(gemm_wsb_kernel, dim3(num_cus()), dim3(512), G::LDS, s, g);
  gemm_prof_end(tok, flops, s);
  return check_launch("hero_wgrad_batch");
}
