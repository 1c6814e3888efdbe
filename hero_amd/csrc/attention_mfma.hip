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
template <int NB, int WPB, int CLS>
__global__ __launch_bounds__(64 * WPB) void attn_mfma_fwd_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  // The pair index is uniform across the wave - readfirstlane says so to the compiler: the sequence bounds (and the dropout
  // seed below) become SCALAR loads issued together, instead of two vector loads each followed by its own s_waitcnt vmcnt(0)
  // in front of every other load of the wave (and a third serial round trip for the seed behind the first MFMAs).
  const int pair = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (pair >= a.S * a.H) return;                       // wave-uniform; no workgroup barriers below
  const int s = pair / a.H, h = pair - s * a.H, D = a.H * 64, ld = 3 * D;
  // packed batches: rows [seq_off[s], seq_off[s+1]); a.L (the maximum) stays the stride of probs / dropout indices
  const int Lm = a.L, Lp = (Lm + 3) & ~3;
  const int row0 = a.seq_off ? a.seq_off[s] : s * Lm;
  const int L = a.seq_off ? a.seq_off[s + 1] - row0 : Lm;
  if (L <= 0) return;
  if ((CLS == 1 && L > 32) || (CLS == 2 && L <= 32)) return;      // wave-uniform: the other launch owns this sequence
  DropCtx drop(a.dropout);
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem) + wave * (32 * NB * RS);
  const bf16_t* qp = static_cast<const bf16_t*>(a.qkv) + (size_t)row0 * ld + h * 64;
  const bf16_t* kp = qp + D;
  const bf16_t* vp = qp + 2 * D;

  stage_tile<NB>(vp, ld, L, Vs, lane);

  // ---- S^T[jt][it] = K Q^T
  f32x16_t sc[NB][NB];
  {
    bf16x8_t kf[NB][4], qf[NB][4];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kf[t][ks] = gfrag(kp, ld, 32 * t + l31, L, ks, half);
        qf[t][ks] = gfrag(qp, ld, 32 * t + l31, L, ks, half);
      }
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int it = 0; it < NB; ++it) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[jt][it][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sc[jt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[jt][ks], qf[it][ks], sc[jt][it], 0, 0, 0);
      }
  }
  // additive key mask of this lane's keys (independent of the query tile)
  float mk[NB][16];
#pragma unroll
  for (int jt = 0; jt < NB; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * jt + acc_row(r, half);
      mk[jt][r] = a.mask ? a.mask[(size_t)s * Lm + min(j, L - 1)] : 0.f;
    }
  // V^T fragments: k-slot e of step ks <-> key 32 jt + 16 ks + 4 half + (e & 3) + 8 (e >> 2), i.e. the
  // keys this lane holds in accumulator registers 8 ks .. 8 ks + 7
  wave_sync_lds();
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
  store_headT<NB>(static_cast<bf16_t*>(a.ctx) + (size_t)row0 * D + h * 64, D, L, cx, lane);
}

template <int NB, int WPB, bool RC, int CLS>       // RC: no saved probabilities - rebuilt from q, k and the saved row statistics
__global__ __launch_bounds__(64 * WPB) void attn_mfma_bwd_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = 32 * NB;
  constexpr int PS = R + 8;                              // [query][key] bf16 row stride (elements)
  constexpr int WAVE_BYTES = 3 * R * RS * 2 + 2 * R * PS * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int pair = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);      // uniform: scalar loads (see the forward kernel)
  if (pair >= a.S * a.H) return;
  const int s = pair / a.H, h = pair - s * a.H, D = a.H * 64, ld = 3 * D;
  const int Lm = a.L, Lp = (Lm + 3) & ~3;
  const int row0 = a.seq_off ? a.seq_off[s] : s * Lm;
  const int L = a.seq_off ? a.seq_off[s + 1] - row0 : Lm;
  if (L <= 0) return;
  if ((CLS == 1 && L > 32) || (CLS == 2 && L <= 32)) return;      // wave-uniform: the other launch owns this sequence
  DropCtx drop(a.dropout);
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem + wave * WAVE_BYTES);
  bf16_t* Qs = Ks + R * RS;
  bf16_t* Os = Qs + R * RS;
  bf16_t* Pl = Os + R * RS;                              // dropped probabilities [i][j]
  bf16_t* Sl = Pl + R * PS;                              // dS [i][j]
  const bf16_t* qp = static_cast<const bf16_t*>(a.qkv) + (size_t)row0 * ld + h * 64;
  const bf16_t* kp = qp + D;
  const bf16_t* vp = qp + 2 * D;
  const bf16_t* op = static_cast<const bf16_t*>(a.dctx) + (size_t)row0 * D + h * 64;

  stage_tile<NB>(kp, ld, L, Ks, lane);
  stage_tile<NB>(qp, ld, L, Qs, lane);
  stage_tile<NB>(op, D, L, Os, lane);

  // ---- dP^T[jt][it] = V dO^T (dP w.r.t. the DROPPED probabilities)
  f32x16_t dp[NB][NB];
  {
    bf16x8_t vf[NB][4];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) vf[t][ks] = gfrag(vp, ld, 32 * t + l31, L, ks, half);
    wave_sync_lds();
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
        for (int ks = 0; ks < 4; ++ks) dp[jt][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jt][ks], of[ks], dp[jt][it], 0, 0, 0);
      }
    }
  }
  // Without saved probabilities: S^T = K Q^T again, from the staged tiles - the same MFMAs on the same operands as the
  // forward's, then the same scale / mask / exp / normalise with the saved row maximum and 1 / row sum: bit-identical P.
  constexpr bool recompute = RC;
  f32x16_t sc[RC ? NB : 1][RC ? NB : 1];
  float mk[RC ? NB : 1][16];
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
#pragma unroll
    for (int jt = 0; jt < NB; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + acc_row(r, half);
        mk[jt][r] = a.mask ? a.mask[(size_t)s * Lm + min(j, L - 1)] : 0.f;
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

  bf16_t* dq = static_cast<bf16_t*>(a.dqkv) + (size_t)row0 * ld + h * 64;
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
      const float2 st = *reinterpret_cast<const float2*>(a.stats + ((size_t)(s * a.H + h) * Lm + min(i, L - 1)) * 2);
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
  store_headT<NB>(dq + D, ld, L, gk, lane);
  store_headT<NB>(dq + 2 * D, ld, L, gv, lane);
}

template <int NB, int WPB, int CLS>
int launch(const HeroAttn& a, bool bwd, hipStream_t s) {
  constexpr int R = 32 * NB;
  const int pairs = a.S * a.H;
  const int grid = (pairs + WPB - 1) / WPB;
  if (bwd) {
    const size_t lds = (size_t)WPB * (3 * R * RS * 2 + 2 * R * (R + 8) * 2);
    if (lds > 65536) {
      HERO_ENSURE_LDS((&attn_mfma_bwd_kernel<NB, WPB, false, CLS>), lds, "attn_mfma_bwd_kernel");
      HERO_ENSURE_LDS((&attn_mfma_bwd_kernel<NB, WPB, true, CLS>), lds, "attn_mfma_bwd_kernel");
    }
    if (a.probs) hipLaunchKernelGGL((attn_mfma_bwd_kernel<NB, WPB, false, CLS>), dim3(grid), dim3(64 * WPB), lds, s, a);
    else hipLaunchKernelGGL((attn_mfma_bwd_kernel<NB, WPB, true, CLS>), dim3(grid), dim3(64 * WPB), lds, s, a);
  } else {
    const size_t lds = (size_t)WPB * R * RS * 2;
    hipLaunchKernelGGL((attn_mfma_fwd_kernel<NB, WPB, CLS>), dim3(grid), dim3(64 * WPB), lds, s, a);
  }
  return check_launch(bwd ? "hero_attention_bwd(mfma)" : "hero_attention_fwd(mfma)");
}

}  // namespace

// bf16, 1 <= L <= 64.  Called by attention.hip's dispatcher.
int attn_mfma_run(const HeroAttn& a, bool bwd, hipStream_t s) {
  if (a.L <= 32) return launch<1, 4, 0>(a, bwd, s);
  if (a.seq_off) {                 // packed: two launches, each taking the sequences of its length class
    const int rc = launch<1, 4, 1>(a, bwd, s);
    return rc ? rc : launch<2, 1, 2>(a, bwd, s);
  }
  return launch<2, 1, 0>(a, bwd, s);
}

}  // namespace hero
