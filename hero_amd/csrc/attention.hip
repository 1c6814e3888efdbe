// Masked multi-head self-attention, head size 64, forward and backward (gfx950): the fp32-VALU
// kernels.  They serve the f32 parity mode at every length and bf16 for 64 < L <= 256; bf16 with
// L <= 64 (every sequence of the TVR step) is dispatched to the matrix-core kernels of
// attention_mfma.hip, which issue ~10x fewer wave instructions per head.
//
// HERO's sequences are short (a subtitle's frames+tokens: 10-40; a clip's frames: <= 100; the
// long-video stress case: 256), so attention is < 1 % of the FLOPs of a layer and is bound by
// launch count and HBM traffic, not by the matrix cores.  One workgroup (4 waves) owns one
// (sequence, head): K and V of that head are staged once in LDS, every wave then walks query rows:
//   scores : a key is owned by LPK adjacent lanes (LPK = 1/2/4 chosen from L so short sequences
//            still fill the wave), each lane dots its 64/LPK slice of q (held in registers)
//            with the K row in LDS, partial dots are combined with wave shuffles;
//   softmax: row max / sum with wave shuffles over the per-wave score row in LDS;
//   P.V    : lane d accumulates output column d over the keys (probability broadcast from LDS,
//            V row read conflict-free).
// The fp32 softmax output is written out for the backward pass (a few MB per layer at HERO's
// lengths), the dropout mask is regenerated from the counter RNG.
#include <stdlib.h>
#include "common.h"

namespace hero {

constexpr int DH = 64;

// LDS written by some lanes of a wave and read by other lanes of the SAME wave: the DS pipe is
// in-order per wave, this only pins the compiler (and its s_waitcnt) at that point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T> struct KvLay;  // LDS row stride (elements) chosen odd in dwords -> conflict-free column walks
template <> struct KvLay<float>  { static constexpr int STRIDE = 65; };
template <> struct KvLay<bf16_t> { static constexpr int STRIDE = 66; };

template <typename T>
__device__ __forceinline__ void stage_head(const T* __restrict__ src, int ld, int L, T* dst, int stride) {
  // src: first row of this head's [L, 64] slice (row stride ld). 16 lanes x 4 elements per row.
  for (int q = threadIdx.x; q < L * 16; q += blockDim.x) {
    const int r = q >> 4, c = (q & 15) * 4;
    const float4 v = V4<T>::ld(src + (size_t)r * ld + c);
    T* d = dst + r * stride + c;
    st1<T>(d + 0, v.x); st1<T>(d + 1, v.y); st1<T>(d + 2, v.z); st1<T>(d + 3, v.w);
  }
}

// dot of the register slice qv[0..DPL) with row `row` of an LDS matrix, columns [p*DPL, (p+1)*DPL)
template <typename T, int DPL> __device__ __forceinline__ float dot_slice(const float (&qv)[DPL], const T* row_ptr);
template <int DPL> __device__ __forceinline__ float dot_slice_f32(const float (&qv)[DPL], const float* rp) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < DPL; ++t) s = fmaf(qv[t], rp[t], s);
  return s;
}
template <int DPL> __device__ __forceinline__ float dot_slice_bf16(const float (&qv)[DPL], const bf16_t* rp) {
  float s = 0.f;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(rp);  // STRIDE 66 and DPL even keep this 4-byte aligned
#pragma unroll
  for (int t = 0; t < DPL / 2; ++t) {
    const uint32_t u = w[t];
    s = fmaf(qv[2 * t], __uint_as_float(u << 16), s);
    s = fmaf(qv[2 * t + 1], __uint_as_float(u & 0xffff0000u), s);
  }
  return s;
}
template <typename T, int DPL> struct Dot;
template <int DPL> struct Dot<float, DPL> {
  static __device__ __forceinline__ float run(const float (&qv)[DPL], const float* rp) { return dot_slice_f32<DPL>(qv, rp); }
};
template <int DPL> struct Dot<bf16_t, DPL> {
  static __device__ __forceinline__ float run(const float (&qv)[DPL], const bf16_t* rp) { return dot_slice_bf16<DPL>(qv, rp); }
};

// load a 64-wide row slice [p*DPL, (p+1)*DPL) from global into registers (same address across the
// lanes that share p -> broadcast loads)
template <typename T, int DPL>
__device__ __forceinline__ void load_row_slice(const T* __restrict__ row, int p, float (&qv)[DPL]) {
#pragma unroll
  for (int t = 0; t < DPL; t += 4) {
    const float4 v = V4<T>::ld(row + p * DPL + t);
    qv[t] = v.x; qv[t + 1] = v.y; qv[t + 2] = v.z; qv[t + 3] = v.w;
  }
}

// same, from an LDS matrix with row stride 64 (lanes sharing p read the same address -> broadcast)
template <typename T, int DPL>
__device__ __forceinline__ void lds_row_slice(const T* row, int p, float (&qv)[DPL]) {
#pragma unroll
  for (int t = 0; t < DPL; ++t) qv[t] = ld1<T>(row + p * DPL + t);
}

template <typename T, int LPK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DPL = DH / LPK, KPP = 64 / LPK, STR = KvLay<T>::STRIDE;
  const int L = a.L, H = a.H, D = H * DH;
  const int s = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Lp = (L + 3) & ~3;
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + (size_t)L * STR;
  float* Ps = reinterpret_cast<float*>(smem + (((size_t)(2 * L * STR) * sizeof(T)) + 15) / 16 * 16) + wave * Lp;

  const T* qkv = static_cast<const T*>(a.qkv) + (size_t)s * L * 3 * D;
  const T* Qg = qkv + h * DH;                  // a query row is read once, by one wave: no LDS copy
  stage_head<T>(qkv + D + h * DH, 3 * D, L, Ks, STR);
  stage_head<T>(qkv + 2 * D + h * DH, 3 * D, L, Vs, STR);
  __syncthreads();

  const float* mask = a.mask ? a.mask + (size_t)s * L : nullptr;
  DropCtx drop(a.dropout);
  const int g = lane / LPK, p = lane % LPK;
  T* ctx = static_cast<T*>(a.ctx) + (size_t)s * L * D + h * DH;
  float* probs = a.probs ? a.probs + ((size_t)(s * H + h) * L) * L : nullptr;

  for (int i = wave; i < L; i += 4) {
    float qv[DPL];
    lds_row_slice<T, DPL>(Qg + (size_t)i * 3 * D, p, qv);
    // ---- scores
    float mx = -3.0e38f;
    for (int j0 = 0; j0 < L; j0 += KPP) {
      const int j = j0 + g;
      float sc = 0.f;
      if (j < L) sc = Dot<T, DPL>::run(qv, Ks + j * STR + p * DPL);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) sc += __shfl_xor(sc, o, 64);
      if (j < L) {
        sc = sc * a.scale + (mask ? mask[j] : 0.f);
        if (p == 0) Ps[j] = sc;
        mx = fmaxf(mx, sc);
      }
    }
    mx = wave_max(mx);
    wave_lds_sync();
    // ---- softmax (each lane owns keys lane, lane+64, ...)
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
      const float e = __expf(Ps[j] - mx);
      Ps[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < L; j += 64) {
      float pr = Ps[j] * inv;
      if (probs) probs[(size_t)i * L + j] = pr;
      if (drop.on()) pr *= drop.mask1(((uint64_t)(s * H + h) * L + i) * (uint64_t)Lp + j);
      Ps[j] = pr;
    }
    wave_lds_sync();
    // ---- context: lane = output column
    float acc = 0.f;
    int j = 0;
    for (; j + 4 <= L; j += 4) {
      const float4 pj = *reinterpret_cast<const float4*>(Ps + j);
      acc = fmaf(pj.x, ld1<T>(Vs + (j + 0) * STR + lane), acc);
      acc = fmaf(pj.y, ld1<T>(Vs + (j + 1) * STR + lane), acc);
      acc = fmaf(pj.z, ld1<T>(Vs + (j + 2) * STR + lane), acc);
      acc = fmaf(pj.w, ld1<T>(Vs + (j + 3) * STR + lane), acc);
    }
    for (; j < L; ++j) acc = fmaf(Ps[j], ld1<T>(Vs + j * STR + lane), acc);
    st1<T>(ctx + (size_t)i * D + lane, acc);
    wave_lds_sync();  // Ps is rewritten by the next row
  }
}

// Backward. LDS: K, V (padded stride), Q, dO (stride 64), and for a chunk of R query rows the
// matrices dS[R][Lp] (already scaled) and Pd[R][Lp] (dropped probabilities).  dK/dV accumulate in
// registers across chunks: wave w owns keys w, w+4, ... (KPW of them per wave).
// QO_GLOBAL (round 3: fp32 up to L = 256, the config-5 length of the parity mode): Q and dO are read from global
// memory (L2) row by row instead of being staged, which leaves the LDS to K, V and the row chunk.
template <typename T, int LPK, int KPW, bool QO_GLOBAL>
__global__ __launch_bounds__(256) void attn_bwd_kernel(HeroAttn a, int R) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DPL = DH / LPK, KPP = 64 / LPK, STR = KvLay<T>::STRIDE;
  constexpr int MAXP = LPK == 1 ? 4 : 1;   // score passes: L <= 256 (LPK 1), <= 32 (LPK 2), <= 16 (LPK 4)
  const int L = a.L, H = a.H, D = H * DH;
  const int s = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Lp = (L + 3) & ~3;
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + (size_t)L * STR;
  const T* qkv = static_cast<const T*>(a.qkv) + (size_t)s * L * 3 * D;
  const T* dctx = static_cast<const T*>(a.dctx) + (size_t)s * L * D + h * DH;
  const T* Qs = QO_GLOBAL ? qkv + h * DH : Vs + (size_t)L * STR;       // row i at Qs + i * QST
  const T* Os = QO_GLOBAL ? dctx : Vs + (size_t)L * STR + (size_t)L * DH;
  const int QST = QO_GLOBAL ? 3 * D : DH, OST = QO_GLOBAL ? D : DH;
  float* dS = reinterpret_cast<float*>(smem + (((size_t)(2 * L * STR + (QO_GLOBAL ? 0 : 2 * L * DH)) * sizeof(T)) + 15) / 16 * 16);
  float* Pd = dS + (size_t)R * Lp;

  if (!QO_GLOBAL) {
    stage_head<T>(qkv + h * DH, 3 * D, L, Vs + (size_t)L * STR, DH);
    stage_head<T>(dctx, D, L, Vs + (size_t)L * STR + (size_t)L * DH, DH);
  }
  stage_head<T>(qkv + D + h * DH, 3 * D, L, Ks, STR);
  stage_head<T>(qkv + 2 * D + h * DH, 3 * D, L, Vs, STR);
  __syncthreads();

  DropCtx drop(a.dropout);
  const int g = lane / LPK, p = lane % LPK;
  const float* probs = a.probs + ((size_t)(s * H + h) * L) * L;
  T* dqkv = static_cast<T*>(a.dqkv) + (size_t)s * L * 3 * D;

  float accK[KPW], accV[KPW];
#pragma unroll
  for (int k = 0; k < KPW; ++k) { accK[k] = 0.f; accV[k] = 0.f; }

  for (int c0 = 0; c0 < L; c0 += R) {
    const int cr = min(R, L - c0);
    // ---- stage this chunk's probability rows (contiguous cr*L floats) into Pd
    {
      const float* src = probs + (size_t)c0 * L;
      for (int q = threadIdx.x; q < cr * L; q += blockDim.x) {
        const int il = q / L, j = q - il * L;
        Pd[(size_t)il * Lp + j] = src[q];
      }
    }
    __syncthreads();
    // ---- phase A: rows of this chunk -> dS, Pd in LDS, dQ to HBM
    for (int il = wave; il < cr; il += 4) {
      const int i = c0 + il;
      float ov[DPL];
      lds_row_slice<T, DPL>(Os + (size_t)i * OST, p, ov);
      float* dSr = dS + (size_t)il * Lp;
      float* Pdr = Pd + (size_t)il * Lp;
      float prs[MAXP], dps[MAXP];
      float delta = 0.f;
#pragma unroll
      for (int ps = 0; ps < MAXP; ++ps) {
        const int j = ps * KPP + g;
        float dp = 0.f;
        prs[ps] = 0.f;
        if (ps * KPP < L) {
          if (j < L) dp = Dot<T, DPL>::run(ov, Vs + j * STR + p * DPL);
#pragma unroll
          for (int o = 1; o < LPK; o <<= 1) dp += __shfl_xor(dp, o, 64);
          if (j < L && p == 0) {
            const float pr = Pdr[j];
            const float m = drop.on() ? drop.mask1(((uint64_t)(s * H + h) * L + i) * (uint64_t)Lp + j) : 1.f;
            dp *= m;                 // dP_ij
            Pdr[j] = pr * m;         // dropped probability (feeds dV)
            prs[ps] = pr;
            delta += dp * pr;
          }
        }
        dps[ps] = dp;
      }
      delta = wave_sum(delta);
#pragma unroll
      for (int ps = 0; ps < MAXP; ++ps) {
        const int j = ps * KPP + g;
        if (ps * KPP < L && j < L && p == 0) dSr[j] = prs[ps] * (dps[ps] - delta) * a.scale;
      }
      wave_lds_sync();
      // dQ[i][lane] = sum_j dS_ij K[j][lane]
      float acc = 0.f;
      for (int j = 0; j < L; ++j) acc = fmaf(dSr[j], ld1<T>(Ks + j * STR + lane), acc);
      st1<T>(dqkv + (size_t)i * 3 * D + h * DH + lane, acc);
    }
    __syncthreads();
    // ---- phase B: keys owned by this wave accumulate over the chunk's rows
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int j = wave + 4 * k;
      if (j < L) {
        float ak = accK[k], av = accV[k];
        for (int il = 0; il < cr; ++il) {
          const int i = c0 + il;
          ak = fmaf(dS[(size_t)il * Lp + j], ld1<T>(Qs + (size_t)i * QST + lane), ak);
          av = fmaf(Pd[(size_t)il * Lp + j], ld1<T>(Os + (size_t)i * OST + lane), av);
        }
        accK[k] = ak; accV[k] = av;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < KPW; ++k) {
    const int j = wave + 4 * k;
    if (j < L) {
      st1<T>(dqkv + (size_t)j * 3 * D + D + h * DH + lane, accK[k]);
      st1<T>(dqkv + (size_t)j * 3 * D + 2 * D + h * DH + lane, accV[k]);
    }
  }
}

// =============================================================================================
// Short sequences (L <= 64: every HERO training shape).  Q, K, V (and dO) of one (sequence, head)
// are staged in LDS as fp32 with 16-byte aligned, conflict-free strides, so the inner loops are
// ds_read_b128 + v_fma only (no bf16 unpacking); all keys fit one pass (64/LPK >= L), softmax runs
// in registers with DPP reductions; the dK/dV phase walks 4 keys at a time (one broadcast b128 of
// dS / P per query row feeds 8 FMAs).
// =============================================================================================
constexpr int KS = 68;   // fp32 row stride of matrices that are read row-per-lane with ds_read_b128

template <typename T>
__device__ __forceinline__ void stage_f32(const T* __restrict__ src, int ld, int L, float* dst, int stride) {
  for (int q = threadIdx.x; q < L * 16; q += blockDim.x) {
    const int r = q >> 4, c = (q & 15) * 4;
    *reinterpret_cast<float4*>(dst + r * stride + c) = V4<T>::ld(src + (size_t)r * ld + c);
  }
}

template <int DPL>
__device__ __forceinline__ void row_regs(const float* row, float (&v)[DPL]) {
#pragma unroll
  for (int t = 0; t < DPL; t += 4) {
    const float4 x = *reinterpret_cast<const float4*>(row + t);
    v[t] = x.x; v[t + 1] = x.y; v[t + 2] = x.z; v[t + 3] = x.w;
  }
}
template <int DPL>
__device__ __forceinline__ float dot_regs(const float (&q)[DPL], const float* row) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < DPL; t += 4) {
    const float4 k = *reinterpret_cast<const float4*>(row + t);
    s0 = fmaf(q[t], k.x, s0); s1 = fmaf(q[t + 1], k.y, s1);
    s0 = fmaf(q[t + 2], k.z, s0); s1 = fmaf(q[t + 3], k.w, s1);
  }
  return s0 + s1;
}

template <typename T, int LPK>
__global__ __launch_bounds__(256) void attn_fwd_small_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DPL = DH / LPK;
  const int H = a.H, D = H * DH;
  const int s = blockIdx.x / H, h = blockIdx.x % H;
  // packed (variable-length) batches: rows [seq_off[s], seq_off[s+1]); a.L is the maximum length and
  // stays the stride of the probabilities and of the dropout indices
  const int Lm = a.L, Lpm = (Lm + 3) & ~3;
  const int row0 = a.seq_off ? a.seq_off[s] : s * Lm;
  const int L = a.seq_off ? a.seq_off[s + 1] - row0 : Lm;
  if (L <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Lp = (L + 3) & ~3;
  float* Qs = reinterpret_cast<float*>(smem);
  float* Ks = Qs + L * DH;
  float* Vs = Ks + L * KS;
  float* Ms = Vs + L * DH;
  float* Ps = Ms + Lp + wave * Lp;

  const T* qkv = static_cast<const T*>(a.qkv) + (size_t)row0 * 3 * D + h * DH;
  stage_f32<T>(qkv, 3 * D, L, Qs, DH);
  stage_f32<T>(qkv + D, 3 * D, L, Ks, KS);
  stage_f32<T>(qkv + 2 * D, 3 * D, L, Vs, DH);
  for (int j = threadIdx.x; j < L; j += 256) Ms[j] = a.mask ? a.mask[(size_t)s * Lm + j] : 0.f;
  __syncthreads();

  DropCtx drop(a.dropout);
  const int g = lane / LPK, p = lane % LPK;
  const bool valid = g < L;
  T* ctx = static_cast<T*>(a.ctx) + (size_t)row0 * D + h * DH;
  float* probs = a.probs ? a.probs + ((size_t)(s * H + h) * Lm) * Lm : nullptr;
  const float mj = valid ? Ms[g] : 0.f;
  const float* krow = Ks + (valid ? g : 0) * KS + p * DPL;

  for (int i = wave; i < L; i += 4) {
    float q[DPL];
    row_regs<DPL>(Qs + i * DH + p * DPL, q);
    float sc = group_sum<LPK>(dot_regs<DPL>(q, krow));
    sc = valid ? sc * a.scale + mj : -3.0e38f;
    const float mx = wave_max(sc);
    const float e = valid ? __expf(sc - mx) : 0.f;
    const float sum = wave_sum(p == 0 ? e : 0.f);
    float pr = e * (1.f / sum);
    if (valid && p == 0) {
      if (probs) probs[(size_t)i * Lm + g] = pr;
      if (drop.on()) pr *= drop.mask1(((uint64_t)(s * H + h) * Lm + i) * (uint64_t)Lpm + g);
      Ps[g] = pr;
    }
    wave_lds_sync();
    float acc0 = 0.f, acc1 = 0.f;
    int j = 0;
    for (; j + 4 <= L; j += 4) {
      const float4 pj = *reinterpret_cast<const float4*>(Ps + j);
      acc0 = fmaf(pj.x, Vs[(j + 0) * DH + lane], acc0);
      acc1 = fmaf(pj.y, Vs[(j + 1) * DH + lane], acc1);
      acc0 = fmaf(pj.z, Vs[(j + 2) * DH + lane], acc0);
      acc1 = fmaf(pj.w, Vs[(j + 3) * DH + lane], acc1);
    }
    for (; j < L; ++j) acc0 = fmaf(Ps[j], Vs[j * DH + lane], acc0);
    st1<T>(ctx + (size_t)i * D + lane, acc0 + acc1);
    wave_lds_sync();
  }
}

template <typename T, int LPK>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(HeroAttn a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DPL = DH / LPK;
  const int H = a.H, D = H * DH;
  const int s = blockIdx.x / H, h = blockIdx.x % H;
  // packed (variable-length) batches: rows [seq_off[s], seq_off[s+1]); a.L is the maximum length and
  // stays the stride of the probabilities and of the dropout indices
  const int Lm = a.L, Lpm = (Lm + 3) & ~3;
  const int row0 = a.seq_off ? a.seq_off[s] : s * Lm;
  const int L = a.seq_off ? a.seq_off[s + 1] - row0 : Lm;
  if (L <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Lp = (L + 3) & ~3;
  float* Qs = reinterpret_cast<float*>(smem);   // [L][64]
  float* Os = Qs + L * DH;                       // [L][64]  dO
  float* Ks = Os + L * DH;                       // [L][68]
  float* Vs = Ks + L * KS;                       // [L][68]
  float* dS = Vs + L * KS;                       // [L][Lp]  (scaled)
  float* Pd = dS + L * Lp;                       // [L][Lp]  P on entry, dropped P afterwards

  const T* qkv = static_cast<const T*>(a.qkv) + (size_t)row0 * 3 * D + h * DH;
  const T* dctx = static_cast<const T*>(a.dctx) + (size_t)row0 * D + h * DH;
  stage_f32<T>(qkv, 3 * D, L, Qs, DH);
  stage_f32<T>(qkv + D, 3 * D, L, Ks, KS);
  stage_f32<T>(qkv + 2 * D, 3 * D, L, Vs, KS);
  stage_f32<T>(dctx, D, L, Os, DH);
  {
    const float* src = a.probs + ((size_t)(s * H + h) * Lm) * Lm;
    for (int q = threadIdx.x; q < L * L; q += 256) {
      const int i = q / L, j = q - i * L;
      Pd[i * Lp + j] = src[(size_t)i * Lm + j];
    }
    for (int q = threadIdx.x; q < L * (Lp - L); q += 256) {   // zero the row padding read by the quad phase
      const int i = q / (Lp - L), j = L + q % (Lp - L);
      Pd[i * Lp + j] = 0.f;
      dS[i * Lp + j] = 0.f;
    }
  }
  __syncthreads();

  DropCtx drop(a.dropout);
  const int g = lane / LPK, p = lane % LPK;
  const bool valid = g < L;
  T* dqkv = static_cast<T*>(a.dqkv) + (size_t)row0 * 3 * D + h * DH;
  const float* vrow = Vs + (valid ? g : 0) * KS + p * DPL;

  // ---- phase A: one wave per query row -> dS row, dropped-P row (LDS), dQ row (HBM)
  for (int i = wave; i < L; i += 4) {
    float ov[DPL];
    row_regs<DPL>(Os + i * DH + p * DPL, ov);
    float dp = group_sum<LPK>(dot_regs<DPL>(ov, vrow));
    float* dSr = dS + i * Lp;
    float* Pdr = Pd + i * Lp;
    float pr = 0.f;
    if (valid) {
      pr = Pdr[g];
      if (drop.on()) {
        const float m = drop.mask1(((uint64_t)(s * H + h) * Lm + i) * (uint64_t)Lpm + g);
        dp *= m;
        if (p == 0) Pdr[g] = pr * m;
      }
    }
    const float delta = wave_sum((valid && p == 0) ? dp * pr : 0.f);
    if (valid && p == 0) dSr[g] = pr * (dp - delta) * a.scale;
    wave_lds_sync();
    float acc0 = 0.f, acc1 = 0.f;
    int j = 0;
    for (; j + 4 <= L; j += 4) {
      const float4 d4 = *reinterpret_cast<const float4*>(dSr + j);
      acc0 = fmaf(d4.x, Ks[(j + 0) * KS + lane], acc0);
      acc1 = fmaf(d4.y, Ks[(j + 1) * KS + lane], acc1);
      acc0 = fmaf(d4.z, Ks[(j + 2) * KS + lane], acc0);
      acc1 = fmaf(d4.w, Ks[(j + 3) * KS + lane], acc1);
    }
    for (; j < L; ++j) acc0 = fmaf(dSr[j], Ks[j * KS + lane], acc0);
    st1<T>(dqkv + (size_t)i * 3 * D + lane, acc0 + acc1);
  }
  __syncthreads();
  // ---- phase B: a wave owns quads of keys; lane = feature column
  const int nquads = Lp >> 2;
  for (int jq = wave; jq < nquads; jq += 4) {
    const int j0 = jq * 4;
    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (int i = 0; i < L; ++i) {
      const float4 d4 = *reinterpret_cast<const float4*>(dS + i * Lp + j0);
      const float4 p4 = *reinterpret_cast<const float4*>(Pd + i * Lp + j0);
      const float qv = Qs[i * DH + lane], ovv = Os[i * DH + lane];
      k0 = fmaf(d4.x, qv, k0); k1 = fmaf(d4.y, qv, k1); k2 = fmaf(d4.z, qv, k2); k3 = fmaf(d4.w, qv, k3);
      v0 = fmaf(p4.x, ovv, v0); v1 = fmaf(p4.y, ovv, v1); v2 = fmaf(p4.z, ovv, v2); v3 = fmaf(p4.w, ovv, v3);
    }
    const float kk[4] = {k0, k1, k2, k3}, vv[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (j0 + t < L) {
        st1<T>(dqkv + (size_t)(j0 + t) * 3 * D + D + lane, kk[t]);
        st1<T>(dqkv + (size_t)(j0 + t) * 3 * D + 2 * D + lane, vv[t]);
      }
  }
}

static size_t small_fwd_lds(int L) { const int Lp = (L + 3) & ~3; return ((size_t)L * (DH + KS + DH) + 5 * Lp) * sizeof(float); }
static size_t small_bwd_lds(int L) { const int Lp = (L + 3) & ~3; return ((size_t)L * (2 * DH + 2 * KS) + 2 * (size_t)L * Lp) * sizeof(float); }

template <typename T, int LPK>
static int launch_small(const HeroAttn& a, bool bwd, hipStream_t s) {
  const size_t lds = bwd ? small_bwd_lds(a.L) : small_fwd_lds(a.L);
  HERO_REQUIRE(lds <= 150 * 1024, "hero_attention(small): L = %d needs %zu bytes of LDS", a.L, lds);
  if (bwd) {
    HERO_ENSURE_LDS((&attn_bwd_small_kernel<T, LPK>), 150 * 1024, "attn_bwd_small_kernel");
    hipLaunchKernelGGL((attn_bwd_small_kernel<T, LPK>), dim3(a.S * a.H), dim3(256), lds, s, a);
  } else {
    HERO_ENSURE_LDS((&attn_fwd_small_kernel<T, LPK>), 150 * 1024, "attn_fwd_small_kernel");
    hipLaunchKernelGGL((attn_fwd_small_kernel<T, LPK>), dim3(a.S * a.H), dim3(256), lds, s, a);
  }
  return check_launch(bwd ? "hero_attention_bwd(small)" : "hero_attention_fwd(small)");
}

constexpr size_t LDS_BUDGET = 150 * 1024;

template <typename T> static size_t fwd_lds(int L) {
  const int Lp = (L + 3) & ~3;
  return (((size_t)(2 * L * KvLay<T>::STRIDE) * sizeof(T)) + 15) / 16 * 16 + (size_t)4 * Lp * sizeof(float);
}
template <typename T> static size_t bwd_lds_fixed(int L, bool qo_global = false) {
  return (((size_t)(2 * L * KvLay<T>::STRIDE + (qo_global ? 0 : 2 * L * DH)) * sizeof(T)) + 15) / 16 * 16;
}
template <typename T> static int max_len(int backward) {
  int L = 1;
  if (!backward) {
    while (L < 1024 && fwd_lds<T>(L + 1) <= LDS_BUDGET) ++L;
    return L;
  }
  while (L < 256 && bwd_lds_fixed<T>(L + 1, true) + (size_t)2 * 4 * ((L + 4) & ~3) * sizeof(float) <= LDS_BUDGET) ++L;
  return L;
}

template <typename T, int LPK>
static int launch_fwd(const HeroAttn& a, hipStream_t s) {
  const size_t lds = fwd_lds<T>(a.L);
  HERO_REQUIRE(lds <= LDS_BUDGET, "hero_attention_fwd: L = %d needs %zu bytes of LDS", a.L, lds);
  HERO_ENSURE_LDS((&attn_fwd_kernel<T, LPK>), LDS_BUDGET, "attn_fwd_kernel");
  hipLaunchKernelGGL((attn_fwd_kernel<T, LPK>), dim3(a.S * a.H), dim3(256), lds, s, a);
  return check_launch("hero_attention_fwd");
}
template <typename T, int LPK, int KPW, bool QO_GLOBAL = false>
static int launch_bwd(const HeroAttn& a, hipStream_t s) {
  const int Lp = (a.L + 3) & ~3;
  const size_t fixed = bwd_lds_fixed<T>(a.L, QO_GLOBAL);
  if (!QO_GLOBAL && fixed + (size_t)2 * 4 * Lp * sizeof(float) > LDS_BUDGET) return launch_bwd<T, LPK, KPW, true>(a, s);
  int R = (int)((LDS_BUDGET - fixed) / ((size_t)2 * Lp * sizeof(float)));
  if (R > a.L) R = a.L;
  R &= ~3;
  if (R < 4) R = a.L < 4 ? a.L : 4;
  const size_t lds = fixed + (size_t)2 * R * Lp * sizeof(float);
  HERO_REQUIRE(lds <= LDS_BUDGET, "hero_attention_bwd: L = %d needs %zu bytes of LDS", a.L, lds);
  HERO_ENSURE_LDS((&attn_bwd_kernel<T, LPK, KPW, QO_GLOBAL>), LDS_BUDGET, "attn_bwd_kernel");
  hipLaunchKernelGGL((attn_bwd_kernel<T, LPK, KPW, QO_GLOBAL>), dim3(a.S * a.H), dim3(256), lds, s, a, R);
  return check_launch("hero_attention_bwd");
}
template <typename T, int LPK>
static int bwd_by_len(const HeroAttn& a, hipStream_t s) {
  if (a.L <= 32) return launch_bwd<T, LPK, 8>(a, s);
  if (a.L <= 64) return launch_bwd<T, LPK, 16>(a, s);
  if (a.L <= 128) return launch_bwd<T, LPK, 32>(a, s);
  return launch_bwd<T, LPK, 64>(a, s);
}
int attn_mfma_run(const HeroAttn& a, bool bwd, hipStream_t s);        // attention_mfma.hip (bf16, L <= 64)
int attn_mfma_long_run(const HeroAttn& a, bool bwd, hipStream_t s);   // attention_mfma_long.hip (bf16, 64 < L <= 256)

template <typename T>
static int run(const HeroAttn& a, bool bwd, hipStream_t s) {
  if (sizeof(T) == 2 && a.L <= 64) return attn_mfma_run(a, bwd, s);
  // longer sequences on the matrix cores too; the backward takes delta_i = dO_i . ctx_i from the forward output
  if (sizeof(T) == 2 && a.L <= 256 && (!bwd || a.ctx)) return attn_mfma_long_run(a, bwd, s);
  if (a.seq_off && a.L > 64) {
    set_error("hero_attention_%s: packed batches with L = %d > 64 need the bf16 matrix-core kernels (and ctx in the backward)",
              bwd ? "bwd" : "fwd", a.L);
    return HERO_ERR_UNSUPPORTED;
  }
  const int lim = max_len<T>(bwd ? 1 : 0);
  if (a.L > lim) {
    set_error("hero_attention_%s: sequence length %d exceeds the LDS-resident limit %d for this dtype", bwd ? "bwd" : "fwd", a.L, lim);
    return HERO_ERR_UNSUPPORTED;
  }
  if (a.L <= 16) return launch_small<T, 4>(a, bwd, s);
  if (a.L <= 32) return launch_small<T, 2>(a, bwd, s);
  if (a.L <= 64) return launch_small<T, 1>(a, bwd, s);
  return bwd ? bwd_by_len<T, 1>(a, s) : launch_fwd<T, 1>(a, s);
}

}  // namespace hero

using namespace hero;

static int check_attn(const HeroAttn* a, bool bwd) {
  HERO_REQUIRE(a && a->qkv, "hero_attention: null qkv");
  HERO_REQUIRE(a->S >= 0 && a->L > 0 && a->H > 0, "hero_attention: bad dims S=%d L=%d H=%d", a->S, a->L, a->H);
  HERO_REQUIRE(a->dtype == HERO_F32 || a->dtype == HERO_BF16, "hero_attention: bad dtype %d", a->dtype);
  HERO_REQUIRE(!a->seq_off || a->L <= (a->dtype == HERO_BF16 ? 256 : 64), "hero_attention: packed batches (seq_off) need L <= %d, got %d",
               a->dtype == HERO_BF16 ? 256 : 64, a->L);
  if (bwd) {
    HERO_REQUIRE((a->probs || a->stats) && a->dctx && a->dqkv, "hero_attention_bwd: probs (or stats)/dctx/dqkv required");
    if (!a->probs)
      HERO_REQUIRE(hero_attention_stats_ok(a->dtype, a->L), "hero_attention_bwd: dtype %d, L = %d needs the saved probabilities (stats alone: bf16, L <= 64)",
                   a->dtype, a->L);
  } else {
    HERO_REQUIRE(a->ctx, "hero_attention_fwd: ctx required");
  }
  return HERO_OK;
}

extern "C" int hero_attention_fwd(const HeroAttn* a, hero_stream_t stream) {
  int rc = check_attn(a, false);
  if (rc) return rc;
  if (a->S == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return a->dtype == HERO_BF16 ? run<bf16_t>(*a, false, s) : run<float>(*a, false, s);
}
extern "C" int hero_attention_bwd(const HeroAttn* a, hero_stream_t stream) {
  int rc = check_attn(a, true);
  if (rc) return rc;
  if (a->S == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return a->dtype == HERO_BF16 ? run<bf16_t>(*a, true, s) : run<float>(*a, true, s);
}
extern "C" int hero_attention_stats_ok(int dtype, int L) { return dtype == HERO_BF16 && L >= 1 && L <= 64 ? 1 : 0; }
extern "C" int hero_attention_max_packed_len(int dtype) { return dtype == HERO_BF16 ? 256 : 64; }
extern "C" int hero_attention_max_len(int dtype, int backward) {
  if (dtype == HERO_BF16) return 256;       // matrix-core kernels (the backward wants a.ctx beyond 64)
  return dtype == HERO_BF16 ? max_len<bf16_t>(backward) : max_len<float>(backward);
}
