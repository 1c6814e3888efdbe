// Masked multi-head self-attention on the matrix cores for 64 < L <= 256 (bf16, head size 64): the
// Temporal Transformer over up to 256 frames (BASELINE configs[4]) and real TVR clips (<= 100 frames).
//
// One WORKGROUP per (sequence, head), one WAVE per 32-row tile; K / V (forward, dQ pass) or Q / dO (dK, dV
// pass) of the head are staged ONCE in LDS ([L][64] bf16, 144-byte rows) and shared by the waves.
// The five products are 32x32x16 bf16 MFMAs in the operand arrangement of attention_mfma.hip
// (S^T = K Q^T with lane <-> query, softmax in the accumulator layout, P / dS fed to the next MFMA
// straight from registers, key- / query-contracted operands by ds_read_b64_tr_b16):
//
//   forward (wave = query tile)   S^T[jt] = K_jt Q^T for all key tiles, softmax over the 32 x L scores in
//                                 registers, probs (fp32, saved for backward) and ctx^T = sum_jt V_jt^T P_jt^T
//   backward, dQ pass (wave = query tile)   delta_i = dO_i . ctx_i;  per key tile: dP^T = V dO^T,
//                                 dS = P (dP M - delta) / 8, dQ^T += K^T dS^T
//   backward, dK/dV pass (wave = key tile)  per query tile: the same dP^T / dS^T tile, P~ and dS go through
//                                 a wave-private 32 x 32 LDS tile and come back transposed:
//                                 dV^T += dO^T P~, dK^T += Q^T dS
// (delta from the forward output needs no pass over the keys: sum_j dP_ij P_ij = dO_i . (P~ V)_i.)
//
// Reference semantics: model/layers.py:129-160; dropout indices / packed batches as in HeroAttn.
#include "attn_mfma.h"

namespace hero {

using namespace attn;

namespace {

// 16 probabilities of query row `prow` at this lane's keys of tile jt (4 runs of 4 consecutive keys)
__device__ __forceinline__ void load_probs(const float* __restrict__ prow, int jt, int half, int L, bool row_ok, bool vec, float (&pr)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j0 = 32 * jt + 8 * q + 4 * half;
    if (vec && j0 + 3 < L) {
      const float4 v = *reinterpret_cast<const float4*>(prow + j0);
      pr[4 * q] = v.x; pr[4 * q + 1] = v.y; pr[4 * q + 2] = v.z; pr[4 * q + 3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) pr[4 * q + e] = prow[min(j0 + e, L - 1)];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) pr[4 * q + e] = (row_ok && j0 + e < L) ? pr[4 * q + e] : 0.f;
  }
}

// dO_i . ctx_i over the 32 head dims [32 half, 32 half + 32) of row i (the caller adds the two halves)
__device__ __forceinline__ float half_row_dot(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, int half) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint4 x = *reinterpret_cast<const uint4*>(a + 32 * half + 8 * c);
    const uint4 y = *reinterpret_cast<const uint4*>(b + 32 * half + 8 * c);
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s = fmaf(__uint_as_float(xs[k] << 16), __uint_as_float(ys[k] << 16), s);
      s = fmaf(__uint_as_float(xs[k] & 0xffff0000u), __uint_as_float(ys[k] & 0xffff0000u), s);
    }
  }
  return s;
}

struct HeadCtx {
  int s, h, D, ld, Lm, Lp, row0, L, njt;
};
__device__ __forceinline__ HeadCtx head_ctx(const HeroAttn& a) {
  HeadCtx c;
  c.s = blockIdx.x / a.H;
  c.h = blockIdx.x - c.s * a.H;
  c.D = a.H * 64;
  c.ld = 3 * c.D;
  c.Lm = a.L;
  c.Lp = (a.L + 3) & ~3;
  c.row0 = a.seq_off ? a.seq_off[c.s] : c.s * a.L;
  c.L = a.seq_off ? a.seq_off[c.s + 1] - c.row0 : a.L;
  c.njt = (c.L + 31) >> 5;
  return c;
}

template <int NB>
__global__ __launch_bounds__(64 * NB) void attn_long_fwd_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const HeadCtx c = head_ctx(a);
  if (c.L <= 0) return;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + 32 * NB * RS;
  const bf16_t* qp = static_cast<const bf16_t*>(a.qkv) + (size_t)c.row0 * c.ld + c.h * 64;
  stage_tile_wg<64 * NB>(qp + c.D, c.ld, c.L, 32 * c.njt, Ks);
  stage_tile_wg<64 * NB>(qp + 2 * c.D, c.ld, c.L, 32 * c.njt, Vs);
  __syncthreads();
  const int it = wave;
  if (it >= c.njt) return;
  const int i = 32 * it + l31;

  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = gfrag(qp, c.ld, i, c.L, ks, half);
  // ---- scores of this query tile against every key tile, scaled + masked
  float p[NB][16];
  float mx = -3.0e38f;
#pragma unroll
  for (int jt = 0; jt < NB; ++jt) {
    if (jt < c.njt) {
      f32x16_t sc;
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lfrag(Ks, 32 * jt + l31, ks, half), qf[ks], sc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * jt + acc_row(r, half);
        const float mk = a.mask ? a.mask[(size_t)c.s * c.Lm + min(j, c.L - 1)] : 0.f;
        const float v = j < c.L ? fmaf(sc[r], a.scale, mk) : -3.0e38f;
        p[jt][r] = v;
        mx = fmaxf(mx, v);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[jt][r] = -3.0e38f;
    }
  }
  mx = fmaxf(mx, xhalf(mx));
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < NB; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * jt + acc_row(r, half);
      const float e = (jt < c.njt && j < c.L) ? __expf(p[jt][r] - mx) : 0.f;
      p[jt][r] = e;
      sum += e;
    }
  sum += xhalf(sum);
  const float inv = 1.f / sum;

  DropCtx drop(a.dropout);
  float* prow = a.probs ? a.probs + ((size_t)(c.s * a.H + c.h) * c.Lm + min(i, c.L - 1)) * c.Lm : nullptr;
  const bool vec = (c.Lm & 3) == 0;
  const uint64_t drow = ((uint64_t)(c.s * a.H + c.h) * c.Lm + i) * (uint64_t)c.Lp;
  f32x16_t cx[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) cx[dt][e] = 0.f;
#pragma unroll
  for (int jt = 0; jt < NB; ++jt) {
    if (jt < c.njt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j0 = 32 * jt + 8 * q + 4 * half;
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (drop.on()) m = drop.mask4((drow + j0) >> 2);
        const float mm[4] = {m.x, m.y, m.z, m.w};
        float pr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pr[e] = p[jt][4 * q + e] * inv;
          p[jt][4 * q + e] = pr[e] * mm[e];
        }
        if (prow && i < c.L) {
          if (vec && j0 + 3 < c.L) {
            *reinterpret_cast<float4*>(prow + j0) = make_float4(pr[0], pr[1], pr[2], pr[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j0 + e < c.L) prow[j0 + e] = pr[e];
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r0 = 32 * jt + 16 * ks + 4 * half;
        const bf16x8_t pk = pack8(&p[jt][8 * ks]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8_t vf = tr_frag(tr_addr(Vs, RS * 2, r0, dt, lane), tr_addr(Vs, RS * 2, r0 + 8, dt, lane));
          cx[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pk, cx[dt], 0, 0, 0);
        }
      }
    }
  }
  store_tileT(static_cast<bf16_t*>(a.ctx) + (size_t)c.row0 * c.D + c.h * 64, c.D, c.L, it, cx, lane);
}

// dS^T tile (lane <-> query i, registers <-> the lane's 16 keys of tile jt) from dP^T, the saved
// probabilities and delta_i; pd = dropped probabilities (operand of dV)
__device__ __forceinline__ void ds_tile(const f32x16_t& dp, const float (&pr)[16], float delta, float scale, const DropCtx& drop,
                                        uint64_t drow, int jt, int half, float (&ds)[16], float (&pd)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j0 = 32 * jt + 8 * q + 4 * half;
    float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
    if (drop.on()) m = drop.mask4((drow + j0) >> 2);
    const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      ds[r] = pr[r] * (dp[r] * mm[e] - delta) * scale;
      pd[r] = pr[r] * mm[e];
    }
  }
}

template <int NB>
__global__ __launch_bounds__(64 * NB) void attn_long_bwd_dq_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const HeadCtx c = head_ctx(a);
  if (c.L <= 0) return;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + 32 * NB * RS;
  const bf16_t* qp = static_cast<const bf16_t*>(a.qkv) + (size_t)c.row0 * c.ld + c.h * 64;
  stage_tile_wg<64 * NB>(qp + c.D, c.ld, c.L, 32 * c.njt, Ks);
  stage_tile_wg<64 * NB>(qp + 2 * c.D, c.ld, c.L, 32 * c.njt, Vs);
  __syncthreads();
  const int it = wave;
  if (it >= c.njt) return;
  const int i = 32 * it + l31, ic = min(i, c.L - 1);
  const bf16_t* op = static_cast<const bf16_t*>(a.dctx) + (size_t)c.row0 * c.D + c.h * 64;
  const bf16_t* cp = static_cast<const bf16_t*>(a.ctx) + (size_t)c.row0 * c.D + c.h * 64;
  bf16x8_t of[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) of[ks] = gfrag(op, c.D, i, c.L, ks, half);
  float delta = half_row_dot(op + (size_t)ic * c.D, cp + (size_t)ic * c.D, half);
  delta += xhalf(delta);

  DropCtx drop(a.dropout);
  const float* prow = a.probs + ((size_t)(c.s * a.H + c.h) * c.Lm + ic) * c.Lm;
  const bool vec = (c.Lm & 3) == 0;
  const uint64_t drow = ((uint64_t)(c.s * a.H + c.h) * c.Lm + i) * (uint64_t)c.Lp;
  f32x16_t gq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) gq[dt][e] = 0.f;
#pragma unroll 1
  for (int jt = 0; jt < c.njt; ++jt) {
    f32x16_t dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lfrag(Vs, 32 * jt + l31, ks, half), of[ks], dp, 0, 0, 0);
    float pr[16], ds[16], pd[16];
    load_probs(prow, jt, half, c.L, i < c.L, vec, pr);
    ds_tile(dp, pr, delta, a.scale, drop, drow, jt, half, ds, pd);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * jt + 16 * ks + 4 * half;
      const bf16x8_t sk = pack8(&ds[8 * ks]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8_t kf = tr_frag(tr_addr(Ks, RS * 2, r0, dt, lane), tr_addr(Ks, RS * 2, r0 + 8, dt, lane));
        gq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, sk, gq[dt], 0, 0, 0);
      }
    }
  }
  store_tileT(static_cast<bf16_t*>(a.dqkv) + (size_t)c.row0 * c.ld + c.h * 64, c.ld, c.L, it, gq, lane);
}

template <int NB>
__global__ __launch_bounds__(64 * NB) void attn_long_bwd_dkv_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PS = 40;                                   // [query][key] bf16 row stride of the wave-private tiles
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const HeadCtx c = head_ctx(a);
  if (c.L <= 0) return;
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Os = Qs + 32 * NB * RS;
  float* dl = reinterpret_cast<float*>(Os + 32 * NB * RS);  // delta[32 NB]
  bf16_t* Pl = reinterpret_cast<bf16_t*>(dl + 32 * NB) + wave * (2 * 32 * PS);
  bf16_t* Sl = Pl + 32 * PS;
  const bf16_t* qp = static_cast<const bf16_t*>(a.qkv) + (size_t)c.row0 * c.ld + c.h * 64;
  const bf16_t* op = static_cast<const bf16_t*>(a.dctx) + (size_t)c.row0 * c.D + c.h * 64;
  const bf16_t* cp = static_cast<const bf16_t*>(a.ctx) + (size_t)c.row0 * c.D + c.h * 64;
  stage_tile_wg<64 * NB>(qp, c.ld, c.L, 32 * c.njt, Qs);
  stage_tile_wg<64 * NB>(op, c.D, c.L, 32 * c.njt, Os);
  {   // delta: two threads per query row
    const int i = threadIdx.x >> 1, hh = threadIdx.x & 1, icl = min(i, c.L - 1);
    float d = half_row_dot(op + (size_t)icl * c.D, cp + (size_t)icl * c.D, hh);
    d += __shfl_xor(d, 1, 64);
    if (hh == 0) dl[i] = i < c.L ? d : 0.f;
  }
  __syncthreads();
  const int jt = wave;
  if (jt >= c.njt) return;
  bf16x8_t vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) vf[ks] = gfrag(qp + 2 * c.D, c.ld, 32 * jt + l31, c.L, ks, half);

  DropCtx drop(a.dropout);
  const bool vec = (c.Lm & 3) == 0;
  f32x16_t gv[2], gk[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { gv[dt][e] = 0.f; gk[dt][e] = 0.f; }
#pragma unroll 1
  for (int it = 0; it < c.njt; ++it) {
    const int i = 32 * it + l31, icl = min(i, c.L - 1);
    f32x16_t dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], lfrag(Os, 32 * it + l31, ks, half), dp, 0, 0, 0);
    const float* prow = a.probs + ((size_t)(c.s * a.H + c.h) * c.Lm + icl) * c.Lm;
    const uint64_t drow = ((uint64_t)(c.s * a.H + c.h) * c.Lm + i) * (uint64_t)c.Lp;
    float pr[16], ds[16], pd[16];
    load_probs(prow, jt, half, c.L, i < c.L, vec, pr);
    ds_tile(dp, pr, dl[i], a.scale, drop, drow, jt, half, ds, pd);
    wave_sync_lds();                                       // the previous tile's transposed reads are done
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      st_bf4(Pl + l31 * PS + 8 * q + 4 * half, pd[4 * q], pd[4 * q + 1], pd[4 * q + 2], pd[4 * q + 3]);
      st_bf4(Sl + l31 * PS + 8 * q + 4 * half, ds[4 * q], ds[4 * q + 1], ds[4 * q + 2], ds[4 * q + 3]);
    }
    wave_sync_lds();
    // contraction over the 32 queries of the tile: k-slot e of step ks <-> query 16 ks + 8 half + e
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 16 * ks + 8 * half;
      const bf16x8_t pf = tr_frag(tr_addr(Pl, PS * 2, r0, 0, lane), tr_addr(Pl, PS * 2, r0 + 4, 0, lane));
      const bf16x8_t sf = tr_frag(tr_addr(Sl, PS * 2, r0, 0, lane), tr_addr(Sl, PS * 2, r0 + 4, 0, lane));
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8_t ot = tr_frag(tr_addr(Os, RS * 2, 32 * it + r0, dt, lane), tr_addr(Os, RS * 2, 32 * it + r0 + 4, dt, lane));
        const bf16x8_t qt = tr_frag(tr_addr(Qs, RS * 2, 32 * it + r0, dt, lane), tr_addr(Qs, RS * 2, 32 * it + r0 + 4, dt, lane));
        gv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, pf, gv[dt], 0, 0, 0);
        gk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, sf, gk[dt], 0, 0, 0);
      }
    }
  }
  bf16_t* dq = static_cast<bf16_t*>(a.dqkv) + (size_t)c.row0 * c.ld + c.h * 64;
  store_tileT(dq + c.D, c.ld, c.L, jt, gk, lane);
  store_tileT(dq + 2 * c.D, c.ld, c.L, jt, gv, lane);
}

template <int NB>
int launch_long(const HeroAttn& a, bool bwd, hipStream_t s) {
  const int grid = a.S * a.H;
  const size_t tiles = (size_t)2 * 32 * NB * RS * 2;
  const size_t lds_kv = tiles + 32 * NB * 4 + (size_t)NB * 2 * 32 * 40 * 2;
  if (tiles > 65536) {
    HERO_ENSURE_LDS((&attn_long_fwd_kernel<NB>), tiles, "attn_long_fwd_kernel");
    HERO_ENSURE_LDS((&attn_long_bwd_dq_kernel<NB>), tiles, "attn_long_bwd_dq_kernel");
  }
  if (lds_kv > 65536) HERO_ENSURE_LDS((&attn_long_bwd_dkv_kernel<NB>), lds_kv, "attn_long_bwd_dkv_kernel");
  if (!bwd) {
    hipLaunchKernelGGL((attn_long_fwd_kernel<NB>), dim3(grid), dim3(64 * NB), tiles, s, a);
    return check_launch("hero_attention_fwd(mfma, long)");
  }
  hipLaunchKernelGGL((attn_long_bwd_dq_kernel<NB>), dim3(grid), dim3(64 * NB), tiles, s, a);
  int rc = check_launch("hero_attention_bwd(mfma, long, dQ)");
  if (rc) return rc;
  hipLaunchKernelGGL((attn_long_bwd_dkv_kernel<NB>), dim3(grid), dim3(64 * NB), lds_kv, s, a);
  return check_launch("hero_attention_bwd(mfma, long, dK dV)");
}

}  // namespace

// bf16, 64 < L <= 256; backward needs the forward output in a.ctx.  Called by attention.hip's dispatcher.
int attn_mfma_long_run(const HeroAttn& a, bool bwd, hipStream_t s) {
  if (a.L <= 128) return launch_long<4>(a, bwd, s);
  return launch_long<8>(a, bwd, s);
}

}  // namespace hero
