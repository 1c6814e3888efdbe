// Short-sequence (L <= 64) masked multi-head self-attention on the matrix cores, bf16, head size 64.
//
// One WAVE owns one (sequence, head).  All five products of attention are 32x32x16 bf16 MFMAs on
// the head's (padded) 32- or 64-row tiles, everything between them stays in registers:
//
//   forward   S^T = K Q^T        A = K rows, B = Q rows: 16-byte fragment loads straight from HBM
//             softmax over keys  in the accumulator layout of S^T: lane <-> query, 16 keys per
//                                lane per tile (+ one cross-half exchange), no LDS
//             ctx^T = V^T P^T    A = V^T by ds_read_b64_tr_b16 from the LDS-staged V tile,
//                                B = P straight from the softmax registers (the key <-> k-slot
//                                assignment is chosen so that each lane already holds its slots)
//   backward  dP^T = V dO^T      same shape as S^T
//             dS                 in registers (lane <-> query), delta_i by the same exchange
//             dQ^T = K^T dS^T    as ctx^T
//             dV^T = dO^T P, dK^T = Q^T dS   contraction over queries: P and dS go through LDS
//                                once ([query][key] bf16) and come back as transposed fragments
//
// The fp32-VALU kernels this replaces (attention.hip, kept for the f32 parity mode and L > 64)
// issue ~5000 wave instructions per head; this one ~400, which moves short-sequence attention
// from VALU-issue-bound to memory/latency-bound.
//
// Reference semantics: model/layers.py:129-160 (scores / sqrt(64) + additive mask, softmax,
// dropout on the probabilities, context).  Dropout indices are those of HeroAttn (hero_hip.h).
#include "attn_mfma.h"

namespace hero {

using namespace attn;

namespace {

// out^T[dt][nt] (head dim x lane-owned row) -> out[row][h*64 + d], rows < L
template <int NB>
__device__ __forceinline__ void store_headT(bf16_t* __restrict__ dst, int ld, int L, const f32x16_t (&acc)[2][NB], int lane) {
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NB; ++nt) {
    const int row = 32 * nt + l31;
    if (row < L) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          st_bf4(dst + (size_t)row * ld + 32 * dt + 8 * q + 4 * half, acc[dt][nt][4 * q], acc[dt][nt][4 * q + 1],
                 acc[dt][nt][4 * q + 2], acc[dt][nt][4 * q + 3]);
    }
  }
}

// CLS (packed batches whose longest sequence needs NB = 2): 0 = every sequence, 1 = only sequences of <= 32 rows (run by
// the NB = 1 kernel), 2 = only the longer ones (NB = 2 kernel).  A ragged TVR batch has 8-48 rows per subtitle: with one
// NB = 2 launch every (sequence, head) pair paid for a 64 x 64 tile at one wave per SIMD - 70 / 110 us per layer forward /
// backward against 16 / 23 us for the same rows at 24 per sequence (profiles/r03_kernel_stats_D2r.csv).
//
// Round 5 - PPW pairs per wave, software-pipelined.  Rounds 2-4 gave every (sequence, head) pair a wave of its own:
// 6144 pairs on 256 CUs x 12-16 resident waves = 1.5 rounds of waves (forward) / 3 rounds (backward, 8 waves per CU
// by LDS), each round one exposed chain scalar loads -> global loads -> LDS -> MFMAs -> stores: 20 / 36 us per launch
// where the bytes need 13.6 / 27 at 5.5 TB/s.  Now a wave walks PPW pairs (pair = wave + k * number of waves, so
// concurrently running waves still touch neighbouring heads of the same rows) and ISSUES the global loads of pair
// k + 1 before it computes pair k: at most two pairs' operands are in registers, the straight-line code lets the
// compiler count vmcnt exactly (VMEM operations of a wave complete in order), and the launch is one full round of waves.
// The arithmetic of a pair is unchanged (same MFMAs on the same operands, same order): results are bit-identical.

// wave-uniform coordinates of a (sequence, head) pair; all of a wave's pairs are resolved at the kernel start (scalar loads
// issued together - resolved later, behind wave-uniform branches, the sequence bounds became vector loads with a wait each)
struct PairCoord { int s, h, row0, L; bool on; };

// Where head h's Q / K / V (and ctx / dctx) columns live.  Product layout: the fused projection's rows [M, 3 * D] (Q | K | V,
// head h at columns h * 64) and [M, D].  -DHERO_ATTN_LAB_HEADMAJOR (tools/lab/attn_layout_ab.py, round 6): per-head PANELS
// [3][H][M][64] / [H][M][64] with M = S * L rows (unpacked launches only) - the layout VERDICT r5 #2 asked to price before
// the QKV GEMM epilogue and the dgrad / wgrad loaders are taught to write / read it.
struct HeadLay { int ld, ldc; size_t q, k, v, c; };
__device__ __forceinline__ HeadLay head_lay(const HeroAttn& a, int h) {
  [[maybe_unused]] const int D = a.H * 64;
#ifdef HERO_ATTN_LAB_HEADMAJOR
  const size_t P = (size_t)a.S * a.L * 64;
  return {64, 64, (size_t)h * P, (size_t)(a.H + h) * P, (size_t)(2 * a.H + h) * P, (size_t)h * P};
#else
  return {3 * D, D, (size_t)h * 64, (size_t)D + h * 64, (size_t)2 * D + h * 64, (size_t)h * 64};
#endif
}
template <int CLS>
__device__ __forceinline__ PairCoord pair_coord(const HeroAttn& a, int pair) {
  const int P = a.S * a.H;
  const int pc = pair < P ? pair : P - 1;                 // past the end: harmless loads of the last pair, nothing computed
  PairCoord c;
  c.s = pc / a.H;
  c.h = pc - c.s * a.H;
  c.row0 = __builtin_amdgcn_readfirstlane(a.seq_off ? a.seq_off[c.s] : c.s * a.L);
  c.L = __builtin_amdgcn_readfirstlane(a.seq_off ? a.seq_off[c.s + 1] - c.row0 : a.L);
  c.on = pair < P && c.L > 0 && !((CLS == 1 && c.L > 32) || (CLS == 2 && c.L <= 32));     // the other launch owns the other class
  return c;
}

// additive key mask of this lane's 16 keys per key tile (accumulator layout: 4 runs of 4 consecutive keys).  Part of a
// pair's load set - loaded inside the compute phase it forced a wait for EVERYTHING older, the next pair's prefetch included
// (VMEM operations complete in order).  Four 16-byte loads when the row stride allows it, else 16 scalar ones.
// M4: rows of a.mask are 16-byte aligned multiples of 4 floats - decided by the LAUNCHER (template parameter): a run-time
// choice between the two load sets is a join, and the compiler made the wave wait for the loads at the join.
template <int NB, bool M4>
__device__ __forceinline__ void load_mask(const HeroAttn& a, const PairCoord& c, float (&mk)[NB][16], int lane) {
  const int half = lane >> 5, Lm = a.L, L = c.L > 0 ? c.L : 1;
  // No mask: the loads still happen (from the start of qkv, always mapped) and a select zeroes the values - an `if (!a.mask)`
  // around the loads is a join of two definitions, and the compiler waits for the loads right there (before the join's copies).
  const bool has = a.mask != nullptr;
  const float* row = has ? a.mask + (size_t)c.s * Lm : reinterpret_cast<const float*>(a.qkv);
  if constexpr (M4) {
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = 32 * jt + 8 * q + 4 * half;          // a run past the row (j0 >= Lm >= L) is never used: any in-row address will do
        const float4 v = *reinterpret_cast<const float4*>(row + (has ? min(j0, Lm - 4) : 0));
        mk[jt][4 * q] = has ? v.x : 0.f; mk[jt][4 * q + 1] = has ? v.y : 0.f; mk[jt][4 * q + 2] = has ? v.z : 0.f; mk[jt][4 * q + 3] = has ? v.w : 0.f;
      }
  } else {
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = row[has ? min(32 * jt + acc_row(r, half), L - 1) : 0];
        mk[jt][r] = has ? v : 0.f;
      }
  }
}

// what a pair's global loads deliver (registers) + its wave-uniform coordinates
template <int NB>
struct FwdIn {
  bf16x8_t kf[NB][4], qf[NB][4];
  uint4 vr[4 * NB];
  float mk[NB][16];
  int s, h, row0, L;
  bool on;
};

template <int NB, bool M4>
__device__ __forceinline__ void fwd_issue(const HeroAttn& a, const PairCoord& pcd, FwdIn<NB>& in, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int s = pcd.s, h = pcd.h;
  const HeadLay hl = head_lay(a, h);
  const int ld = hl.ld;
  const int row0 = pcd.row0, L = pcd.L;
  in.s = s; in.h = h; in.row0 = row0; in.L = L; in.on = pcd.on;
  // A pair this launch does not own (the other length class of a packed batch, or past the end) is not computed; its loads
  // stay in the instruction stream - a wave-uniform branch around them made the compiler wait vmcnt(0) behind every issue,
  // i.e. no prefetch at all - but all go to row 0 of the tensor (one cache line per operand), selected without a branch.
  const int Lc = pcd.on ? L : 1;
  const bf16_t* rowp = static_cast<const bf16_t*>(a.qkv) + (size_t)(pcd.on ? row0 : 0) * ld;
  const bf16_t* qp = rowp + hl.q;
  const bf16_t* kp = rowp + hl.k;
  const bf16_t* vp = rowp + hl.v;
  const int c = (lane & 7) * 8;
#pragma unroll
  for (int it = 0; it < 4 * NB; ++it) {
    const int r = it * 8 + (lane >> 3);
    in.vr[it] = *reinterpret_cast<const uint4*>(vp + (size_t)min(r, Lc - 1) * ld + c);
  }
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      in.kf[t][ks] = gfrag(kp, ld, 32 * t + l31, Lc, ks, half);
      in.qf[t][ks] = gfrag(qp, ld, 32 * t + l31, Lc, ks, half);
    }
  load_mask<NB, M4>(a, pcd, in.mk, lane);
}

// V rows -> the wave's LDS tile [32 NB][RS], rows >= L zeroed (stage_tile's second half)
template <int NB>
__device__ __forceinline__ void fwd_stage(const FwdIn<NB>& in, bf16_t* Vs, int lane) {
  const int c = (lane & 7) * 8;
#pragma unroll
  for (int it = 0; it < 4 * NB; ++it) {
    const int r = it * 8 + (lane >> 3);
    uint4 t = in.vr[it];
    if (r >= in.L) t = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(Vs + r * RS + c) = t;
  }
}

template <int NB>
__device__ __forceinline__ void fwd_compute(const HeroAttn& a, const FwdIn<NB>& in, const bf16_t* Vs, const DropCtx& drop, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int s = in.s, h = in.h, row0 = in.row0, L = in.L;
  const int Lm = a.L, Lp = (Lm + 3) & ~3;
  // ---- S^T[jt][it] = K Q^T
  f32x16_t sc[NB][NB];
#pragma unroll
  for (int jt = 0; jt < NB; ++jt)
#pragma unroll
    for (int it = 0; it < NB; ++it) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[jt][it][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sc[jt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(in.kf[jt][ks], in.qf[it][ks], sc[jt][it], 0, 0, 0);
    }
  const float (&mk)[NB][16] = in.mk;                   // additive key mask of this lane's keys (loaded with the pair's operands)
  // V^T fragments: k-slot e of step ks <-> key 32 jt + 16 ks + 4 half + (e & 3) + 8 (e >> 2), i.e. the
  // keys this lane holds in accumulator registers 8 ks .. 8 ks + 7
  bf16x8_t vf[2][NB][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r0 = 32 * jt + 16 * ks + 4 * half;
        vf[dt][jt][ks] = tr_frag(tr_addr(Vs, RS * 2, r0, dt, lane), tr_addr(Vs, RS * 2, r0 + 8, dt, lane));
      }

  f32x16_t cx[2][NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int i = 32 * it + l31;
    float p[NB][16];
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + acc_row(r, half);
        const float v = j < L ? fmaf(sc[jt][it][r], a.scale, mk[jt][r]) : -3.0e38f;
        p[jt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, xhalf(mx));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + acc_row(r, half);
        const float e = j < L ? __expf(p[jt][r] - mx) : 0.f;
        p[jt][r] = e;
        sum += e;
      }
    sum += xhalf(sum);
    const float inv = 1.f / sum;
    if (a.stats && half == 0 && i < L)                 // what the backward needs to rebuild this row of P from q, k
      *reinterpret_cast<float2*>(a.stats + ((size_t)(s * a.H + h) * Lm + i) * 2) = make_float2(mx, inv);
    float* prow = a.probs ? a.probs + ((size_t)(s * a.H + h) * Lm + min(i, L - 1)) * Lm : nullptr;
    const uint64_t drow = ((uint64_t)(s * a.H + h) * Lm + i) * (uint64_t)Lp;
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = 32 * jt + 8 * q + 4 * half;
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (drop.on()) m = drop.mask4((drow + j0) >> 2);
        const float mm[4] = {m.x, m.y, m.z, m.w};
        float pr4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pr4[e] = p[jt][4 * q + e] * inv;
          p[jt][4 * q + e] = pr4[e] * mm[e];
        }
        if (prow && i < L) {
          if ((Lm & 3) == 0 && j0 + 3 < L) {
            *reinterpret_cast<float4*>(prow + j0) = make_float4(pr4[0], pr4[1], pr4[2], pr4[3]);   // one 16-byte store per run of 4 columns
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j0 + e < L) prow[j0 + e] = pr4[e];
          }
        }
      }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) cx[dt][it][e] = 0.f;
#pragma unroll
      for (int jt = 0; jt < NB; ++jt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          cx[dt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][jt][ks], pack8(&p[jt][8 * ks]), cx[dt][it], 0, 0, 0);
    }
  }
  {
    const HeadLay hl = head_lay(a, h);
    store_headT<NB>(static_cast<bf16_t*>(a.ctx) + (size_t)row0 * hl.ldc + hl.c, hl.ldc, L, cx, lane);
  }
}

template <int NB, int WPB, int CLS, int PPW, bool M4>       // second bound: waves per SIMD the two-pair kernel must fit (<= 168 registers)
__global__ __launch_bounds__(64 * WPB, (PPW == 2 ? 3 : 1)) void attn_mfma_fwd_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The wave index is uniform - readfirstlane says so to the compiler: the sequence bounds (and the dropout seed) become
  // SCALAR loads issued together, instead of vector loads each followed by its own s_waitcnt vmcnt(0).
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;
  if (wid >= a.S * a.H) return;                        // wave-uniform; no workgroup barriers below
  DropCtx drop(a.dropout);
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem) + wave * (32 * NB * RS);
  // two named register sets, explicit steps (an array of structs indexed by the unrolled loop counter ended up in scratch)
  const PairCoord c0 = pair_coord<CLS>(a, wid), c1 = pair_coord<CLS>(a, PPW > 1 ? wid + nw : wid),
                  c2 = pair_coord<CLS>(a, PPW > 2 ? wid + 2 * nw : wid);
  FwdIn<NB> A, B;
#ifdef HERO_ATTN_LATE_ISSUE          // lab (tools/lab/attn_ab.py): the next pair's loads are issued once this pair's have landed
  constexpr bool LATE = true;
#else
  constexpr bool LATE = false;
#endif
  fwd_issue<NB, M4>(a, c0, A, lane);
  if (PPW > 1 && !LATE) fwd_issue<NB, M4>(a, c1, B, lane);                 // next pair's loads fly during this pair
  fwd_stage<NB>(A, Vs, lane);
  wave_sync_lds();
  if (PPW > 1 && LATE) fwd_issue<NB, M4>(a, c1, B, lane);
  if (A.on) fwd_compute<NB>(a, A, Vs, drop, lane);
  if (PPW > 1) {
    if (PPW > 2) fwd_issue<NB, M4>(a, c2, A, lane);
    wave_sync_lds();                                   // the previous pair's transpose reads of the tile are done
    fwd_stage<NB>(B, Vs, lane);
    wave_sync_lds();
    if (B.on) fwd_compute<NB>(a, B, Vs, drop, lane);
  }
  if (PPW > 2) {
    wave_sync_lds();
    fwd_stage<NB>(A, Vs, lane);
    wave_sync_lds();
    if (A.on) fwd_compute<NB>(a, A, Vs, drop, lane);
  }
}

template <int NB>
struct BwdIn {
  uint4 kr[4 * NB], qr[4 * NB], orw[4 * NB];           // K, Q, dO rows as loaded (16 B per lane per 8 rows)
  bf16x8_t vf[NB][4];                                  // V row fragments, straight from global memory
  float mk[NB][16];                                    // RC: additive key mask of this lane's keys
  float2 st[NB];                                       // RC: saved row maximum and 1 / row sum of this lane's query rows
  int s, h, row0, L;
  bool on;
};

template <int NB, bool RC, bool M4>
__device__ __forceinline__ void bwd_issue(const HeroAttn& a, const PairCoord& pcd, BwdIn<NB>& in, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int s = pcd.s, h = pcd.h;
  const HeadLay hl = head_lay(a, h);
  const int ld = hl.ld, D = hl.ldc;
  const int row0 = pcd.row0, L = pcd.L;
  in.s = s; in.h = h; in.row0 = row0; in.L = L; in.on = pcd.on;
  const int Lc = pcd.on ? L : 1;                         // not owned: every load goes to row 0 (see fwd_issue)
  const bf16_t* rowp = static_cast<const bf16_t*>(a.qkv) + (size_t)(pcd.on ? row0 : 0) * ld;
  const bf16_t* qp = rowp + hl.q;
  const bf16_t* kp = rowp + hl.k;
  const bf16_t* vp = rowp + hl.v;
  const bf16_t* op = static_cast<const bf16_t*>(a.dctx) + (size_t)(pcd.on ? row0 : 0) * D + hl.c;
  const int c = (lane & 7) * 8;
#pragma unroll
  for (int it = 0; it < 4 * NB; ++it) {
    const int r = min(it * 8 + (lane >> 3), Lc - 1);
    in.kr[it] = *reinterpret_cast<const uint4*>(kp + (size_t)r * ld + c);
    in.qr[it] = *reinterpret_cast<const uint4*>(qp + (size_t)r * ld + c);
    in.orw[it] = *reinterpret_cast<const uint4*>(op + (size_t)r * D + c);
  }
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) in.vf[t][ks] = gfrag(vp, ld, 32 * t + l31, Lc, ks, half);
  if constexpr (RC) {
    load_mask<NB, M4>(a, pcd, in.mk, lane);
#pragma unroll
    for (int it = 0; it < NB; ++it)
      in.st[it] = *reinterpret_cast<const float2*>(a.stats + ((size_t)(s * a.H + h) * a.L + min(32 * it + l31, Lc - 1)) * 2);
  }
}

template <int NB>
__device__ __forceinline__ void bwd_stage(const BwdIn<NB>& in, bf16_t* Ks, bf16_t* Qs, bf16_t* Os, int lane) {
  const int c = (lane & 7) * 8;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int it = 0; it < 4 * NB; ++it) {
    const int r = it * 8 + (lane >> 3);
    uint4 k4 = in.kr[it], q4 = in.qr[it], o4 = in.orw[it];     // by value: a select between two uint4 objects takes their
    if (r >= in.L) { k4 = z; q4 = z; o4 = z; }                 // addresses and parks the whole register set in scratch
    *reinterpret_cast<uint4*>(Ks + r * RS + c) = k4;
    *reinterpret_cast<uint4*>(Qs + r * RS + c) = q4;
    *reinterpret_cast<uint4*>(Os + r * RS + c) = o4;
  }
}

template <int NB, bool RC>       // RC: no saved probabilities - rebuilt from q, k and the saved row statistics
__device__ __forceinline__ void bwd_compute(const HeroAttn& a, const BwdIn<NB>& in, bf16_t* Ks, const DropCtx& drop, int lane) {
  constexpr int R = 32 * NB;
  constexpr int PS = R + 8;                              // [query][key] bf16 row stride (elements)
  const int half = lane >> 5, l31 = lane & 31;
  const int s = in.s, h = in.h, row0 = in.row0, L = in.L;
  const HeadLay hl = head_lay(a, h);
  const int ld = hl.ld;
  const int Lm = a.L, Lp = (Lm + 3) & ~3;
  bf16_t* Qs = Ks + R * RS;
  bf16_t* Os = Qs + R * RS;
  bf16_t* Pl = Os + R * RS;                              // dropped probabilities [i][j]
  bf16_t* Sl = Pl + R * PS;                              // dS [i][j]

  // ---- dP^T[jt][it] = V dO^T (dP w.r.t. the DROPPED probabilities)
  f32x16_t dp[NB][NB];
  {
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      bf16x8_t of[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) of[ks] = lfrag(Os, 32 * it + l31, ks, half);
#pragma unroll
      for (int jt = 0; jt < NB; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[jt][it][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp[jt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(in.vf[jt][ks], of[ks], dp[jt][it], 0, 0, 0);
      }
    }
  }
  // Without saved probabilities: S^T = K Q^T again, from the staged tiles - the same MFMAs on the same operands as the
  // forward's, then the same scale / mask / exp / normalise with the saved row maximum and 1 / row sum: bit-identical P.
  constexpr bool recompute = RC;
  f32x16_t sc[RC ? NB : 1][RC ? NB : 1];
  const float (&mk)[NB][16] = in.mk;
  if constexpr (RC) {
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      bf16x8_t qf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[ks] = lfrag(Qs, 32 * it + l31, ks, half);
#pragma unroll
      for (int jt = 0; jt < NB; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[jt][it][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          sc[jt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lfrag(Ks, 32 * jt + l31, ks, half), qf[ks], sc[jt][it], 0, 0, 0);
      }
    }
  }
  // K^T fragments for dQ (same key <-> k-slot assignment as the forward's V^T)
  bf16x8_t kf[2][NB][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r0 = 32 * jt + 16 * ks + 4 * half;
        kf[dt][jt][ks] = tr_frag(tr_addr(Ks, RS * 2, r0, dt, lane), tr_addr(Ks, RS * 2, r0 + 8, dt, lane));
      }

  bf16_t* dqrow = static_cast<bf16_t*>(a.dqkv) + (size_t)row0 * ld;
  bf16_t* dq = dqrow + hl.q;
  f32x16_t gq[2][NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int i = 32 * it + l31;
    const float* prow = recompute ? nullptr : a.probs + ((size_t)(s * a.H + h) * Lm + min(i, L - 1)) * Lm;
    const uint64_t drow = ((uint64_t)(s * a.H + h) * Lm + i) * (uint64_t)Lp;
    float pr[NB][16], ds[NB][16];
    // The saved probabilities of this lane's 16 accumulator slots are 4 runs of 4 consecutive columns: four 16-byte
    // loads when the row stride allows it, used UNCONDITIONALLY (masked by a multiplication).  Written as
    // `(i < L && j < L) ? prow[j] : 0` the compiler sank each of the 16 scalar loads into its own conditional block,
    // each followed by s_waitcnt vmcnt(0): 16 serial round trips per 32-row block.
    const float rowok = i < L ? 1.f : 0.f;
    if constexpr (RC) {
      const float2 st = in.st[it];
#pragma unroll
      for (int jt = 0; jt < NB; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * jt + acc_row(r, half);
          const float v = j < L ? fmaf(sc[jt][it][r], a.scale, mk[jt][r]) : -3.0e38f;
          const float e = j < L ? __expf(v - st.x) : 0.f;
          pr[jt][r] = e * st.y * rowok;
        }
    } else if ((Lm & 3) == 0) {
#pragma unroll
      for (int jt = 0; jt < NB; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j0 = 32 * jt + 8 * q + 4 * half;
          const float4 v = *reinterpret_cast<const float4*>(prow + min(j0, Lm - 4));
          pr[jt][4 * q + 0] = v.x * (j0 + 0 < L ? rowok : 0.f);
          pr[jt][4 * q + 1] = v.y * (j0 + 1 < L ? rowok : 0.f);
          pr[jt][4 * q + 2] = v.z * (j0 + 2 < L ? rowok : 0.f);
          pr[jt][4 * q + 3] = v.w * (j0 + 3 < L ? rowok : 0.f);
        }
    } else {
#pragma unroll
      for (int jt = 0; jt < NB; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * jt + acc_row(r, half);
          pr[jt][r] = prow[min(j, L - 1)] * (j < L ? rowok : 0.f);
        }
    }
    float delta = 0.f;
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = 32 * jt + 8 * q + 4 * half;
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (drop.on()) m = drop.mask4((drow + j0) >> 2);
        const float mm[4] = {m.x, m.y, m.z, m.w};
        float pd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = dp[jt][it][4 * q + e] * mm[e];       // dP w.r.t. the softmax output
          ds[jt][4 * q + e] = g;
          delta = fmaf(g, pr[jt][4 * q + e], delta);
          pd[e] = pr[jt][4 * q + e] * mm[e];
        }
        st_bf4(Pl + i * PS + j0, pd[0], pd[1], pd[2], pd[3]);
      }
    delta += xhalf(delta);
#pragma unroll
    for (int jt = 0; jt < NB; ++jt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[jt][r] = pr[jt][r] * (ds[jt][r] - delta) * a.scale;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        st_bf4(Sl + i * PS + 32 * jt + 8 * q + 4 * half, ds[jt][4 * q], ds[jt][4 * q + 1], ds[jt][4 * q + 2], ds[jt][4 * q + 3]);
    }
    // dQ^T[dt][it] = K^T dS^T, B straight from the registers
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) gq[dt][it][e] = 0.f;
#pragma unroll
      for (int jt = 0; jt < NB; ++jt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          gq[dt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[dt][jt][ks], pack8(&ds[jt][8 * ks]), gq[dt][it], 0, 0, 0);
    }
  }
  store_headT<NB>(dq, ld, L, gq, lane);

  // ---- dV^T = dO^T P_dropped, dK^T = Q^T dS: contraction over the queries, k-slot e of step ks <-> query
  //      32 it + 16 ks + 8 half + e for both operands (two transpose reads of 4 rows each)
  wave_sync_lds();
  f32x16_t gv[2][NB], gk[2][NB];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { gv[dt][jt][e] = 0.f; gk[dt][jt][e] = 0.f; }
#pragma unroll
  for (int it = 0; it < NB; ++it)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * it + 16 * ks + 8 * half;
      bf16x8_t of[2], qf[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        of[dt] = tr_frag(tr_addr(Os, RS * 2, r0, dt, lane), tr_addr(Os, RS * 2, r0 + 4, dt, lane));
        qf[dt] = tr_frag(tr_addr(Qs, RS * 2, r0, dt, lane), tr_addr(Qs, RS * 2, r0 + 4, dt, lane));
      }
#pragma unroll
      for (int jt = 0; jt < NB; ++jt) {
        const bf16x8_t pf = tr_frag(tr_addr(Pl, PS * 2, r0, jt, lane), tr_addr(Pl, PS * 2, r0 + 4, jt, lane));
        const bf16x8_t sf = tr_frag(tr_addr(Sl, PS * 2, r0, jt, lane), tr_addr(Sl, PS * 2, r0 + 4, jt, lane));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          gv[dt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[dt], pf, gv[dt][jt], 0, 0, 0);
          gk[dt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[dt], sf, gk[dt][jt], 0, 0, 0);
        }
      }
    }
  store_headT<NB>(dqrow + hl.k, ld, L, gk, lane);
  store_headT<NB>(dqrow + hl.v, ld, L, gv, lane);
}

template <int NB, int WPB, bool RC, int CLS, int PPW, bool M4>       // second bound: at least two waves per SIMD (<= 256 registers) for the multi-pair kernels
__global__ __launch_bounds__(64 * WPB, (PPW > 1 ? 2 : 1)) void attn_mfma_bwd_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = 32 * NB;
  constexpr int WAVE_BYTES = 3 * R * RS * 2 + 2 * R * (R + 8) * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;      // uniform: scalar loads (see the forward kernel)
  if (wid >= a.S * a.H) return;
  DropCtx drop(a.dropout);
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem + wave * WAVE_BYTES);
  const PairCoord c0 = pair_coord<CLS>(a, wid), c1 = pair_coord<CLS>(a, PPW > 1 ? wid + nw : wid),
                  c2 = pair_coord<CLS>(a, PPW > 2 ? wid + 2 * nw : wid);
  BwdIn<NB> A, B;
  bf16_t* Qs = Ks + R * RS;
  bf16_t* Os = Qs + R * RS;
#ifdef HERO_ATTN_LATE_ISSUE
  constexpr bool LATE = true;
#else
  constexpr bool LATE = false;
#endif
  bwd_issue<NB, RC, M4>(a, c0, A, lane);
  if (PPW > 1 && !LATE) bwd_issue<NB, RC, M4>(a, c1, B, lane);             // next pair's loads fly during this pair
  bwd_stage<NB>(A, Ks, Qs, Os, lane);
  wave_sync_lds();
  if (PPW > 1 && LATE) bwd_issue<NB, RC, M4>(a, c1, B, lane);
  if (A.on) bwd_compute<NB, RC>(a, A, Ks, drop, lane);
  if (PPW > 1) {
    if (PPW > 2) bwd_issue<NB, RC, M4>(a, c2, A, lane);
    wave_sync_lds();                                   // the previous pair's reads of the tiles are done
    bwd_stage<NB>(B, Ks, Qs, Os, lane);
    wave_sync_lds();
    if (B.on) bwd_compute<NB, RC>(a, B, Ks, drop, lane);
  }
  if (PPW > 2) {
    wave_sync_lds();
    bwd_stage<NB>(A, Ks, Qs, Os, lane);
    wave_sync_lds();
    if (A.on) bwd_compute<NB, RC>(a, A, Ks, drop, lane);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 64-row class (32 < L <= 64), round 5: TWO waves per (sequence, head) pair.  The one-wave kernels above hold a 64 x 64
// problem in one wave: 270-340 registers = one wave per SIMD, four pairs in flight per CU, 28 / 50 us per layer forward /
// backward on the ragged TVR batch (profiles/r04_kernel_stats_D2r.csv) - a third of its sequences.  Here wave w of a pair
// owns QUERY tile w for everything that is per query (S^T, softmax, ctx / dP, dS, dQ) and KEY tile w for the two products
// that contract over the queries (dK, dV); K / Q / dO / V are staged once per pair, half the rows by each wave; workgroup
// barriers separate the phases.  Same MFMAs on the same operands in the same order as the one-wave kernels: bit-identical
// results.  One pair per workgroup (128 threads); the backward's P tile reuses the K tile's LDS once both waves have read it
// (37 KB per pair: four pairs = eight waves per CU).
// ------------------------------------------------------------------------------------------------------------------
template <int CLS, bool M4>
__global__ __launch_bounds__(128, 2) void attn_mfma_fwd2_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
  const PairCoord pc = pair_coord<CLS>(a, __builtin_amdgcn_readfirstlane(blockIdx.x));
  if (!pc.on) return;                                   // uniform across the WORKGROUP (both waves share the pair): no barrier is skipped by one wave only
  const int s = pc.s, h = pc.h, row0 = pc.row0, L = pc.L, D = a.H * 64, ld = 3 * D;
  const int Lm = a.L, Lp = (Lm + 3) & ~3;
  DropCtx drop(a.dropout);
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem);
  const bf16_t* qp = static_cast<const bf16_t*>(a.qkv) + (size_t)row0 * ld + h * 64;
  const bf16_t* kp = qp + D;
  const bf16_t* vp = qp + 2 * D;
  // ---- loads: this wave's half of the V rows, K fragments of both key tiles, Q fragments of its query tile, the key mask
  uint4 vr[4];
  const int c = (lane & 7) * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) vr[it] = *reinterpret_cast<const uint4*>(vp + (size_t)min(32 * w + it * 8 + (lane >> 3), L - 1) * ld + c);
  bf16x8_t kf[2][4], qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[0][ks] = gfrag(kp, ld, l31, L, ks, half);
    kf[1][ks] = gfrag(kp, ld, 32 + l31, L, ks, half);
    qf[ks] = gfrag(qp, ld, 32 * w + l31, L, ks, half);
  }
  float mk[2][16];
  load_mask<2, M4>(a, pc, mk, lane);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = 32 * w + it * 8 + (lane >> 3);
    uint4 t = vr[it];
    if (r >= L) t = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(Vs + r * RS + c) = t;
  }
  // ---- S^T[jt] = K_jt Q_w^T
  f32x16_t sc[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[jt][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[jt][ks], qf[ks], sc[jt], 0, 0, 0);
  }
  __syncthreads();                                       // the V tile is staged (both halves)
  bf16x8_t vf[2][2][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r0 = 32 * jt + 16 * ks + 4 * half;
        vf[dt][jt][ks] = tr_frag(tr_addr(Vs, RS * 2, r0, dt, lane), tr_addr(Vs, RS * 2, r0 + 8, dt, lane));
      }
  const int i = 32 * w + l31;
  float p[2][16];
  float mx = -3.0e38f;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * jt + acc_row(r, half);
      const float v = j < L ? fmaf(sc[jt][r], a.scale, mk[jt][r]) : -3.0e38f;
      p[jt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = fmaxf(mx, xhalf(mx));
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * jt + acc_row(r, half);
      const float e = j < L ? __expf(p[jt][r] - mx) : 0.f;
      p[jt][r] = e;
      sum += e;
    }
  sum += xhalf(sum);
  const float inv = 1.f / sum;
  if (a.stats && half == 0 && i < L)
    *reinterpret_cast<float2*>(a.stats + ((size_t)(s * a.H + h) * Lm + i) * 2) = make_float2(mx, inv);
  float* prow = a.probs ? a.probs + ((size_t)(s * a.H + h) * Lm + min(i, L - 1)) * Lm : nullptr;
  const uint64_t drow = ((uint64_t)(s * a.H + h) * Lm + i) * (uint64_t)Lp;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j0 = 32 * jt + 8 * q + 4 * half;
      float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
      if (drop.on()) m = drop.mask4((drow + j0) >> 2);
      const float mm[4] = {m.x, m.y, m.z, m.w};
      float pr4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pr4[e] = p[jt][4 * q + e] * inv;
        p[jt][4 * q + e] = pr4[e] * mm[e];
      }
      if (prow && i < L) {
        if ((Lm & 3) == 0 && j0 + 3 < L) {
          *reinterpret_cast<float4*>(prow + j0) = make_float4(pr4[0], pr4[1], pr4[2], pr4[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j0 + e < L) prow[j0 + e] = pr4[e];
        }
      }
    }
  f32x16_t cx[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) cx[dt][e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        cx[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][jt][ks], pack8(&p[jt][8 * ks]), cx[dt], 0, 0, 0);
  }
  store_tileT(static_cast<bf16_t*>(a.ctx) + (size_t)row0 * D + h * 64, D, L, w, cx, lane);
}

template <bool RC, int CLS, bool M4>
__global__ __launch_bounds__(128, 2) void attn_mfma_bwd2_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = 64, PS = RS;                          // P / dS tiles [query][key] share the head tiles' row stride
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, l31 = lane & 31;
  const PairCoord pc = pair_coord<CLS>(a, __builtin_amdgcn_readfirstlane(blockIdx.x));
  if (!pc.on) return;                                   // uniform across the workgroup
  const int s = pc.s, h = pc.h, row0 = pc.row0, L = pc.L, D = a.H * 64, ld = 3 * D;
  const int Lm = a.L, Lp = (Lm + 3) & ~3;
  DropCtx drop(a.dropout);
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Qs = Ks + R * RS;
  bf16_t* Os = Qs + R * RS;
  bf16_t* Sl = Os + R * RS;                              // dS [i][j]
  bf16_t* Pl = Ks;                                       // dropped probabilities [i][j]: the K tile's LDS, after barrier B
  const bf16_t* qp = static_cast<const bf16_t*>(a.qkv) + (size_t)row0 * ld + h * 64;
  const bf16_t* kp = qp + D;
  const bf16_t* vp = qp + 2 * D;
  const bf16_t* op = static_cast<const bf16_t*>(a.dctx) + (size_t)row0 * D + h * 64;
  // ---- loads: this wave's half of the K / Q / dO rows, V fragments of both key tiles, mask, row statistics
  uint4 kr[4], qr[4], orw[4];
  const int c = (lane & 7) * 8;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = min(32 * w + it * 8 + (lane >> 3), L - 1);
    kr[it] = *reinterpret_cast<const uint4*>(kp + (size_t)r * ld + c);
    qr[it] = *reinterpret_cast<const uint4*>(qp + (size_t)r * ld + c);
    orw[it] = *reinterpret_cast<const uint4*>(op + (size_t)r * D + c);
  }
  bf16x8_t vf[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vf[t][ks] = gfrag(vp, ld, 32 * t + l31, L, ks, half);
  const int i = 32 * w + l31;
  float mk[2][16];
  float2 st = make_float2(0.f, 0.f);
  if constexpr (RC) {
    load_mask<2, M4>(a, pc, mk, lane);
    st = *reinterpret_cast<const float2*>(a.stats + ((size_t)(s * a.H + h) * Lm + min(i, L - 1)) * 2);
  }
  {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = 32 * w + it * 8 + (lane >> 3);
      uint4 k4 = kr[it], q4 = qr[it], o4 = orw[it];
      if (r >= L) { k4 = z; q4 = z; o4 = z; }
      *reinterpret_cast<uint4*>(Ks + r * RS + c) = k4;
      *reinterpret_cast<uint4*>(Qs + r * RS + c) = q4;
      *reinterpret_cast<uint4*>(Os + r * RS + c) = o4;
    }
  }
  __syncthreads();                                       // A: K, Q, dO staged (both halves)
  // ---- phase 1, query tile w: dP^T[jt] = V_jt dO_w^T, (RC) S^T[jt] = K_jt Q_w^T
  f32x16_t dp[2];
  {
    bf16x8_t of[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) of[ks] = lfrag(Os, i, ks, half);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) dp[jt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dp[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jt][ks], of[ks], dp[jt], 0, 0, 0);
    }
  }
  f32x16_t sc[2];
  if constexpr (RC) {
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = lfrag(Qs, i, ks, half);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[jt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        sc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lfrag(Ks, 32 * jt + l31, ks, half), qf[ks], sc[jt], 0, 0, 0);
    }
  }
  // K^T fragments for dQ (same key <-> k-slot assignment as the forward's V^T)
  bf16x8_t kf[2][2][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r0 = 32 * jt + 16 * ks + 4 * half;
        kf[dt][jt][ks] = tr_frag(tr_addr(Ks, RS * 2, r0, dt, lane), tr_addr(Ks, RS * 2, r0 + 8, dt, lane));
      }
  __syncthreads();                                       // B: both waves are done with the K tile - P may overwrite it
  bf16_t* dq = static_cast<bf16_t*>(a.dqkv) + (size_t)row0 * ld + h * 64;
  {
    const float* prow = RC ? nullptr : a.probs + ((size_t)(s * a.H + h) * Lm + min(i, L - 1)) * Lm;
    const uint64_t drow = ((uint64_t)(s * a.H + h) * Lm + i) * (uint64_t)Lp;
    float pr[2][16], ds[2][16];
    const float rowok = i < L ? 1.f : 0.f;
    if constexpr (RC) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * jt + acc_row(r, half);
          const float v = j < L ? fmaf(sc[jt][r], a.scale, mk[jt][r]) : -3.0e38f;
          const float e = j < L ? __expf(v - st.x) : 0.f;
          pr[jt][r] = e * st.y * rowok;
        }
    } else if ((Lm & 3) == 0) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j0 = 32 * jt + 8 * q + 4 * half;
          const float4 v = *reinterpret_cast<const float4*>(prow + min(j0, Lm - 4));
          pr[jt][4 * q + 0] = v.x * (j0 + 0 < L ? rowok : 0.f);
          pr[jt][4 * q + 1] = v.y * (j0 + 1 < L ? rowok : 0.f);
          pr[jt][4 * q + 2] = v.z * (j0 + 2 < L ? rowok : 0.f);
          pr[jt][4 * q + 3] = v.w * (j0 + 3 < L ? rowok : 0.f);
        }
    } else {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * jt + acc_row(r, half);
          pr[jt][r] = prow[min(j, L - 1)] * (j < L ? rowok : 0.f);
        }
    }
    float delta = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = 32 * jt + 8 * q + 4 * half;
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (drop.on()) m = drop.mask4((drow + j0) >> 2);
        const float mm[4] = {m.x, m.y, m.z, m.w};
        float pd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = dp[jt][4 * q + e] * mm[e];
          ds[jt][4 * q + e] = g;
          delta = fmaf(g, pr[jt][4 * q + e], delta);
          pd[e] = pr[jt][4 * q + e] * mm[e];
        }
        st_bf4(Pl + i * PS + j0, pd[0], pd[1], pd[2], pd[3]);
      }
    delta += xhalf(delta);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[jt][r] = pr[jt][r] * (ds[jt][r] - delta) * a.scale;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        st_bf4(Sl + i * PS + 32 * jt + 8 * q + 4 * half, ds[jt][4 * q], ds[jt][4 * q + 1], ds[jt][4 * q + 2], ds[jt][4 * q + 3]);
    }
    f32x16_t gq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) gq[dt][e] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          gq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[dt][jt][ks], pack8(&ds[jt][8 * ks]), gq[dt], 0, 0, 0);
    }
    store_tileT(dq, ld, L, w, gq, lane);
  }
  __syncthreads();                                       // C: P and dS of all 64 queries are in the LDS
  // ---- phase 2, key tile w: dV^T = dO^T P_dropped, dK^T = Q^T dS over the 64 queries
  f32x16_t gv[2], gk[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { gv[dt][e] = 0.f; gk[dt][e] = 0.f; }
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * it + 16 * ks + 8 * half;
      bf16x8_t of[2], qf[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        of[dt] = tr_frag(tr_addr(Os, RS * 2, r0, dt, lane), tr_addr(Os, RS * 2, r0 + 4, dt, lane));
        qf[dt] = tr_frag(tr_addr(Qs, RS * 2, r0, dt, lane), tr_addr(Qs, RS * 2, r0 + 4, dt, lane));
      }
      const bf16x8_t pf = tr_frag(tr_addr(Pl, PS * 2, r0, w, lane), tr_addr(Pl, PS * 2, r0 + 4, w, lane));
      const bf16x8_t sf = tr_frag(tr_addr(Sl, PS * 2, r0, w, lane), tr_addr(Sl, PS * 2, r0 + 4, w, lane));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[dt], pf, gv[dt], 0, 0, 0);
        gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[dt], sf, gk[dt], 0, 0, 0);
      }
    }
  store_tileT(dq + D, ld, L, w, gk, lane);
  store_tileT(dq + 2 * D, ld, L, w, gv, lane);
}

static bool mask_by_fours(const HeroAttn& a) { return !a.mask || ((a.L & 3) == 0 && ((uintptr_t)a.mask & 15) == 0); }

template <int CLS, bool M4>
int launch2m(const HeroAttn& a, bool bwd, hipStream_t s) {
  const int pairs = a.S * a.H;
  if (bwd) {
    const size_t lds = (size_t)4 * 64 * RS * 2;          // K (then P), Q, dO, dS tiles: 72 KiB, above the 64 KiB default
    HERO_ENSURE_LDS((&attn_mfma_bwd2_kernel<false, CLS, M4>), lds, "attn_mfma_bwd2_kernel");
    HERO_ENSURE_LDS((&attn_mfma_bwd2_kernel<true, CLS, M4>), lds, "attn_mfma_bwd2_kernel");
    if (a.probs) hipLaunchKernelGGL((attn_mfma_bwd2_kernel<false, CLS, M4>), dim3(pairs), dim3(128), lds, s, a);
    else hipLaunchKernelGGL((attn_mfma_bwd2_kernel<true, CLS, M4>), dim3(pairs), dim3(128), lds, s, a);
  } else {
    hipLaunchKernelGGL((attn_mfma_fwd2_kernel<CLS, M4>), dim3(pairs), dim3(128), (size_t)64 * RS * 2, s, a);
  }
  return check_launch(bwd ? "hero_attention_bwd(mfma, 64 rows)" : "hero_attention_fwd(mfma, 64 rows)");
}
template <int CLS>
int launch2(const HeroAttn& a, bool bwd, hipStream_t s) {
  return mask_by_fours(a) ? launch2m<CLS, true>(a, bwd, s) : launch2m<CLS, false>(a, bwd, s);
}

static int cu_count() {
  int dev = 0, v = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
  return v > 0 ? v : 256;
}

template <int NB, int WPB, int CLS, int PPW, bool M4>
int launch_ppw(const HeroAttn& a, bool bwd, hipStream_t s) {
  constexpr int R = 32 * NB;
  const int pairs = a.S * a.H;
  const int waves = (pairs + PPW - 1) / PPW;
  const int grid = (waves + WPB - 1) / WPB;
  if (bwd) {
    const size_t lds = (size_t)WPB * (3 * R * RS * 2 + 2 * R * (R + 8) * 2);
    if (lds > 65536) {
      HERO_ENSURE_LDS((&attn_mfma_bwd_kernel<NB, WPB, false, CLS, PPW, M4>), lds, "attn_mfma_bwd_kernel");
      HERO_ENSURE_LDS((&attn_mfma_bwd_kernel<NB, WPB, true, CLS, PPW, M4>), lds, "attn_mfma_bwd_kernel");
    }
    if (a.probs) hipLaunchKernelGGL((attn_mfma_bwd_kernel<NB, WPB, false, CLS, PPW, M4>), dim3(grid), dim3(64 * WPB), lds, s, a);
    else hipLaunchKernelGGL((attn_mfma_bwd_kernel<NB, WPB, true, CLS, PPW, M4>), dim3(grid), dim3(64 * WPB), lds, s, a);
  } else {
    const size_t lds = (size_t)WPB * R * RS * 2;
    hipLaunchKernelGGL((attn_mfma_fwd_kernel<NB, WPB, CLS, PPW, M4>), dim3(grid), dim3(64 * WPB), lds, s, a);
  }
  return check_launch(bwd ? "hero_attention_bwd(mfma)" : "hero_attention_fwd(mfma)");
}

// Pairs per wave of the BACKWARD: as many as make the launch ONE round of resident waves - 8 per CU (two per SIMD: <= 256
// registers, and its 19 KB of LDS tiles per wave allow no more) - at most 3; small launches keep one pair per wave.  The bench
// batch: 6144 pairs = 3 x 2048.  The forward keeps one pair per wave (measurements below).
static int g_force_ppw = 0;          // hero_attention_force_ppw: 0 = heuristic, 1..3 = pairs per wave (tuning hook / tests)

template <int WPB, int CLS, bool M4>
int launch_m(const HeroAttn& a, bool bwd, hipStream_t s) {
  const long slots = (long)cu_count() * (bwd ? 8 : 12);
  const long pairs = (long)a.S * a.H;
  // Measured on MI355X (tools/lab/attn_ab.py, the bench batch's 5760 pairs, hipGraph timing): backward 32.5 / 31.2 / 31.0 us
  // with 1 / 2 / 3 pairs per wave (round 4: 35.7), forward 17.6 / 18.6 / 18.7 (round 4: 17.9) - the forward's pairs are short
  // enough that a second round of independently scheduled waves overlaps loads and compute better than a wave's own
  // prefetch does, so it keeps one pair per wave; the hook can still force 2 or 3.
  // Length-class launches of a packed batch (CLS != 0) own only some of the pairs they walk: with several pairs per wave the
  // waves that drew two or three owned pairs become the tail (ragged TVR batch, backward: 47 us with three pairs per wave on a
  // box where one pair per wave took ~40) - they keep one pair per wave as well.
  const int ppw = g_force_ppw ? g_force_ppw : ((!bwd || CLS != 0) ? 1 : (pairs > 2 * slots ? 3 : (pairs > slots ? 2 : 1)));
  if (ppw == 3) return launch_ppw<1, WPB, CLS, 3, M4>(a, bwd, s);
  if (ppw == 2) return launch_ppw<1, WPB, CLS, 2, M4>(a, bwd, s);
  return launch_ppw<1, WPB, CLS, 1, M4>(a, bwd, s);
}
template <int WPB, int CLS>
int launch(const HeroAttn& a, bool bwd, hipStream_t s) {
  return mask_by_fours(a) ? launch_m<WPB, CLS, true>(a, bwd, s) : launch_m<WPB, CLS, false>(a, bwd, s);
}

}  // namespace

// bf16, 1 <= L <= 64.  Called by attention.hip's dispatcher.
int attn_mfma_run(const HeroAttn& a, bool bwd, hipStream_t s) {
  if (a.L <= 32) return launch<4, 0>(a, bwd, s);
  if (a.seq_off) {                 // packed: two launches, each taking the sequences of its length class
    const int rc = launch<4, 1>(a, bwd, s);
    return rc ? rc : launch2<2>(a, bwd, s);
  }
  return launch2<0>(a, bwd, s);
}

}  // namespace hero

// Tuning hook (like hero_gemm_force_config): pairs per wave of the L <= 32 matrix-core attention kernels; 0 = heuristic.
extern "C" int hero_attention_force_ppw(int ppw) {
  if (ppw < 0 || ppw > 3) { hero::set_error("hero_attention_force_ppw: 0 (heuristic) .. 3"); return HERO_ERR_ARG; }
  hero::g_force_ppw = ppw;
  return HERO_OK;
}
