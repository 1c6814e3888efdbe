// VSM / VCMR task-head kernels (gfx950): query pooling, row normalisation, masked max of the
// query x frame scores, in-batch ranking loss, start/end localisation loss - forward and backward.
//
// These are the "small ops" of model/pretrain.py:62-292 and model/encoder.py:460-471: a few MB of
// fp32 data per step, but ~250 PyTorch launches of 2-10 us each when written as tensor ops.  Each
// stage is one kernel here, sized so that a workgroup owns one query / one video / one row and
// needs no inter-workgroup communication (the only atomics are the 32-way parameter-gradient
// accumulations).  All of it is HBM/latency-bound integer-free fp32 work: no MFMA, wave-level DPP
// reductions, 16-byte accesses.
#include "common.h"

namespace hero {
namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads, result to all
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// ------------------------------------------------------------------------------------------------
// A. query pooling
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void query_pool_fwd_kernel(HeroQueryPool a) {
  extern __shared__ float sc[];                      // [L]
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* q = static_cast<const T*>(a.q) + (size_t)b * a.L * a.D;
  for (int l = wave; l < a.L; l += 4) {
    float s = 0.f;
    for (int d = lane * 4; d < a.D; d += 256) s += dot4(V4<T>::ld(q + (size_t)l * a.D + d), *reinterpret_cast<const float4*>(a.w + d));
    s = wave_sum(s);
    if (lane == 0) {
      const float m = a.mask[(size_t)b * a.L + l];
      sc[l] = s * m + (1.f - m) * -10000.f;          // mask_logits, model/modeling_utils.py:42-43
    }
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -3.0e38f;
    for (int l = lane; l < a.L; l += 64) mx = fmaxf(mx, sc[l]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int l = lane; l < a.L; l += 64) sum += __expf(sc[l] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int l = lane; l < a.L; l += 64) {
      const float p = __expf(sc[l] - mx) * inv;
      sc[l] = p;
      a.att[(size_t)b * a.L + l] = p;
    }
  }
  __syncthreads();
  for (int d = threadIdx.x * 4; d < a.D; d += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < a.L; ++l) {
      const float p = sc[l];
      const float4 v = V4<T>::ld(q + (size_t)l * a.D + d);
      acc.x = fmaf(p, v.x, acc.x); acc.y = fmaf(p, v.y, acc.y); acc.z = fmaf(p, v.z, acc.z); acc.w = fmaf(p, v.w, acc.w);
    }
    *reinterpret_cast<float4*>(a.pooled + (size_t)b * a.D + d) = acc;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void query_pool_bwd_kernel(HeroQueryPool a) {
  extern __shared__ float sm[];                      // da[L], ds[L]
  float* da = sm;
  float* ds = sm + a.L;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* q = static_cast<const T*>(a.q) + (size_t)b * a.L * a.D;
  const float* dp = a.dpooled + (size_t)b * a.D;
  const float* att = a.att + (size_t)b * a.L;
  for (int l = wave; l < a.L; l += 4) {              // d att[l] = <dpooled, q[l]>
    float s = 0.f;
    for (int d = lane * 4; d < a.D; d += 256) s += dot4(V4<T>::ld(q + (size_t)l * a.D + d), *reinterpret_cast<const float4*>(dp + d));
    s = wave_sum(s);
    if (lane == 0) da[l] = s;
  }
  __syncthreads();
  if (wave == 0) {
    float dot = 0.f;
    for (int l = lane; l < a.L; l += 64) dot += att[l] * da[l];
    dot = wave_sum(dot);
    for (int l = lane; l < a.L; l += 64) ds[l] = att[l] * (da[l] - dot) * a.mask[(size_t)b * a.L + l];
  }
  __syncthreads();
  T* dq = static_cast<T*>(a.dq) + (size_t)b * a.L * a.D;
  for (int d = threadIdx.x * 4; d < a.D; d += 1024) {
    const float4 g = *reinterpret_cast<const float4*>(dp + d);
    const float4 w = *reinterpret_cast<const float4*>(a.w + d);
    float4 dw = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < a.L; ++l) {
      const float p = att[l], s = ds[l];
      const float4 v = V4<T>::ld(q + (size_t)l * a.D + d);
      V4<T>::st(dq + (size_t)l * a.D + d, make_float4(fmaf(p, g.x, s * w.x), fmaf(p, g.y, s * w.y), fmaf(p, g.z, s * w.z), fmaf(p, g.w, s * w.w)));
      dw.x = fmaf(s, v.x, dw.x); dw.y = fmaf(s, v.y, dw.y); dw.z = fmaf(s, v.z, dw.z); dw.w = fmaf(s, v.w, dw.w);
    }
    // dw: [B, D] partial sums, one row per query (round 4; the caller folds them in a fixed order - round 3 added the B
    // shares to one [D] vector with fp32 atomics)
    if (a.dw) *reinterpret_cast<float4*>(a.dw + (size_t)b * a.D + d) = dw;
  }
}

// ------------------------------------------------------------------------------------------------
// B. row normalisation: one wave per row
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(HeroRowNorm a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  const T* x = static_cast<const T*>(a.x) + (size_t)row * a.cols;
  float ss = 0.f;
  for (int c = lane * 4; c < a.cols; c += 256) { const float4 v = V4<T>::ld(x + c); ss += dot4(v, v); }
  ss = wave_sum(ss);
  const float n = sqrtf(ss);
  const bool clamped = n < a.eps;
  const float r = 1.f / (clamped ? a.eps : n);
  if (lane == 0) a.rnorm[row] = clamped ? -r : r;
  float* y = a.y + (size_t)row * a.cols;
  for (int c = lane * 4; c < a.cols; c += 256) {
    float4 v = V4<T>::ld(x + c);
    v.x *= r; v.y *= r; v.z *= r; v.w *= r;
    *reinterpret_cast<float4*>(y + c) = v;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(HeroRowNorm a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  const T* x = static_cast<const T*>(a.x) + (size_t)row * a.cols;
  const float* dy = a.dy + (size_t)row * a.cols;
  const float rs = a.rnorm[row], r = fabsf(rs);
  float dot = 0.f;                                    // <y, dy> with y = x*r
  if (rs > 0.f) {
    for (int c = lane * 4; c < a.cols; c += 256) dot += dot4(V4<T>::ld(x + c), *reinterpret_cast<const float4*>(dy + c));
    dot = wave_sum(dot) * r;
  }
  T* dx = static_cast<T*>(a.dx) + (size_t)row * a.cols;
  const float k = dot * r;                            // dx = r*(dy - y*dot) = r*dy - x*(r*r*dot)
  for (int c = lane * 4; c < a.cols; c += 256) {
    const float4 v = V4<T>::ld(x + c), g = *reinterpret_cast<const float4*>(dy + c);
    V4<T>::st(dx + c, make_float4(r * (g.x - v.x * k), r * (g.y - v.y * k), r * (g.z - v.z * k), r * (g.w - v.w * k)));
  }
}

// ------------------------------------------------------------------------------------------------
// C. mask_logits + max over the frames of a video
// ------------------------------------------------------------------------------------------------
// Round 6: one WAVE per (query, video) pair, a lane per frame (coalesced; the round-5 kernel gave every pair ONE thread that walked
// its 60 frames with a load each: 13 us for 1024 pairs).  arg = the FIRST frame that attains the maximum, as a serial `>` walk finds it.
__global__ __launch_bounds__(256) void score_max_fwd_kernel(HeroScoreMax a) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= a.M * a.N) return;
  const int m = i / a.N, n = i - m * a.N;
  const float* s = a.s + (size_t)m * a.ld_s + (size_t)n * a.L;
  const float* mk = a.mask + (size_t)n * a.L;
  float best = -3.0e38f;
  int arg = 0x7fffffff;
  for (int l = lane; l < a.L; l += 64) {
    const float k = mk[l];
    const float v = s[l] * k + (1.f - k) * -10000.f;
    if (v > best) { best = v; arg = l; }             // within a lane the frames come in increasing order: first maximum kept
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  if (lane == 0) {
    a.out[i] = best;
    a.arg[i] = arg == 0x7fffffff ? 0 : arg;
  }
}
// dqn: one workgroup per query m.  Round 6: the (frame, weight) pair of every video is resolved FIRST (arg -> mask -> weight is a
// dependent chain: walked inside the accumulation loop it cost one memory round trip per video, 20 us for 32 videos), staged in
// the LDS, and the accumulation loop then issues its row loads four videos at a time.  Same terms, same order (a zero weight adds
// g * v = 0 exactly, rows are finite): bit-identical to the round-5 kernel.
__global__ __launch_bounds__(256) void score_max_bwd_q_kernel(HeroScoreMax a) {
  extern __shared__ __attribute__((aligned(16))) char smem_q[];
  float* sg = reinterpret_cast<float*>(smem_q);
  int* sl = reinterpret_cast<int*>(smem_q) + a.N;
  const int m = blockIdx.x;
  const float gc = a.gc[0] * a.gc_scale, gq = a.gq[0] * a.gq_scale;
  for (int n = threadIdx.x; n < a.N; n += 256) {
    const int i = m * a.N + n, l = a.arg[i];
    sl[n] = l;
    sg[n] = (gc * a.ds_ctx[i] + gq * a.ds_q[i]) * a.mask[(size_t)n * a.L + l];
  }
  __syncthreads();
  for (int d = threadIdx.x * 4; d < a.D; d += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int n = 0;
    for (; n + 4 <= a.N; n += 4) {
      float4 v[4];
      float g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = sg[n + u];
        v[u] = *reinterpret_cast<const float4*>(a.cn + ((size_t)(n + u) * a.L + sl[n + u]) * a.D + d);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc.x = fmaf(g[u], v[u].x, acc.x); acc.y = fmaf(g[u], v[u].y, acc.y); acc.z = fmaf(g[u], v[u].z, acc.z); acc.w = fmaf(g[u], v[u].w, acc.w);
      }
    }
    for (; n < a.N; ++n) {
      const float g = sg[n];
      const float4 v = *reinterpret_cast<const float4*>(a.cn + ((size_t)n * a.L + sl[n]) * a.D + d);
      acc.x = fmaf(g, v.x, acc.x); acc.y = fmaf(g, v.y, acc.y); acc.z = fmaf(g, v.z, acc.z); acc.w = fmaf(g, v.w, acc.w);
    }
    *reinterpret_cast<float4*>(a.dqn + (size_t)m * a.D + d) = acc;
  }
}
// dcn: one workgroup per OWN video n; it owns rows [n*L, (n+1)*L) of dcn, so no atomics
__global__ __launch_bounds__(256) void score_max_bwd_c_kernel(HeroScoreMax a) {
  const int nl = blockIdx.x, n = a.n0 + nl;
  float* out = a.dcn + (size_t)nl * a.L * a.D;
  for (size_t i = threadIdx.x * 4; i < (size_t)a.L * a.D; i += 1024) *reinterpret_cast<float4*>(out + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const float gc = a.gc[0] * a.gc_scale, gq = a.gq[0] * a.gq_scale;
  // each thread owns its columns in every row of this video: plain read-modify-write, fixed order
  for (int d = threadIdx.x * 4; d < a.D; d += 1024) {
    for (int m = 0; m < a.M; ++m) {
      const int i = m * a.N + n, l = a.arg[i];
      const float g = (gc * a.ds_ctx[i] + gq * a.ds_q[i]) * a.mask[(size_t)n * a.L + l];
      if (g != 0.f) {
        const float4 v = *reinterpret_cast<const float4*>(a.qn + (size_t)m * a.D + d);
        float4* p = reinterpret_cast<float4*>(out + (size_t)l * a.D + d);
        float4 o = *p;
        o.x = fmaf(g, v.x, o.x); o.y = fmaf(g, v.y, o.y); o.z = fmaf(g, v.z, o.z); o.w = fmaf(g, v.w, o.w);
        *p = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// D. ranking loss over all in-batch negatives
// ------------------------------------------------------------------------------------------------
// (a kernel, not hipMemsetAsync: inside a captured hipGraph every producer / consumer of a buffer
// should be a kernel node of the same chain)
__global__ void zero_f32_kernel(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

__device__ __forceinline__ void rank_term(const HeroRankLoss& a, float pos, float neg, float& l, float& g) {
  if (a.lse) {
    const float z = neg - pos;
    l = z > 15.f ? z + log1pf(__expf(-z)) : log1pf(__expf(z));
    g = 1.f / (1.f + __expf(-z));
  } else {
    l = fmaxf(a.margin + neg - pos, 0.f);
    g = l > 0.f ? 1.f : 0.f;
  }
}
// blockIdx.y = 0: row m against the other videos (l_ctx); 1: positive query m against the queries
// of other videos in the column of its own video (l_q)
__global__ __launch_bounds__(256) void rank_loss_kernel(HeroRankLoss a) {
  __shared__ float red[4];
  const int m = blockIdx.x, per = a.nq / a.nv, own = m / per;
  const bool qside = blockIdx.y == 1;
  const float pos = a.s[(size_t)m * a.nv + own];
  const int cnt = qside ? a.nq : a.nv;               // candidates along the row / column
  const int nneg = qside ? a.nq - per : a.nv - 1;
  float* ds = qside ? a.ds_q : a.ds_ctx;
  const float scale = 1.f / ((float)nneg * (float)a.nq);          // d mean_rows(mean_negs) / d term
  float loss = 0.f, dpos = 0.f;
  for (int c = threadIdx.x; c < cnt; c += 256) {
    const bool is_pos = qside ? (c / per == own) : (c == own);
    if (is_pos) continue;
    const float neg = qside ? a.s[(size_t)c * a.nv + own] : a.s[(size_t)m * a.nv + c];
    float w = 1.f;
    if (a.hard) {                                     // rank among this row's / column's negatives
      int rank = 0;
      for (int c2 = 0; c2 < cnt; ++c2) {
        const bool p2 = qside ? (c2 / per == own) : (c2 == own);
        if (p2) continue;
        const float v2 = qside ? a.s[(size_t)c2 * a.nv + own] : a.s[(size_t)m * a.nv + c2];
        rank += (v2 > neg || (v2 == neg && c2 < c)) ? 1 : 0;
      }
      w = rank < a.pool ? a.hard_w : a.easy_w;
    }
    float l, g;
    rank_term(a, pos, neg, l, g);
    loss += w * l;
    dpos -= w * g;
    if (!qside) {
      ds[(size_t)m * a.nv + c] = w * g * scale;
    } else if (m % per == 0) {
      // the `per` positive queries of a video share its column: the first of them writes the cell, adding the `per`
      // terms in a fixed order (round 3: every positive's workgroup added its own term with an fp32 atomic)
      float tot = 0.f;
      for (int j = 0; j < per; ++j) {
        float lj, gj;
        rank_term(a, a.s[(size_t)(own * per + j) * a.nv + own], neg, lj, gj);
        tot += w * gj * scale;
      }
      ds[(size_t)c * a.nv + own] = tot;
    }
  }
  loss = block_sum(loss, red);
  dpos = block_sum(dpos, red);
  if (threadIdx.x == 0) {
    (qside ? a.loss_q_rows : a.loss_ctx_rows)[m] = loss / (float)nneg;
    ds[(size_t)m * a.nv + own] = dpos * scale;       // the positive's own cell: one writer on either side
  }
}

// ------------------------------------------------------------------------------------------------
// E. start / end localisation: one workgroup per (query, video) pair b
// ------------------------------------------------------------------------------------------------
constexpr int MAXK = 15;

template <typename T>
__global__ __launch_bounds__(256) void st_ed_fwd_kernel(HeroStEd a) {
  extern __shared__ float sm[];                      // sim[L], lg[2][L]
  __shared__ float red[4];
  float* sim = sm;
  float* lg = sm + a.L;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* ctx = static_cast<const T*>(a.ctx) + (size_t)b * a.L * a.D;
  const float* q2 = a.q2 + (size_t)b * a.D;
  // Round 6: four rows per trip, their loads issued together (a row per trip - load, wave sum, store behind a branch - was one
  // memory round trip per row: 15 serial ones per wave for 60 frames, most of the kernel's 22 us).  Same sums, same order.
  for (int l0 = wave; l0 < a.L; l0 += 16) {
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = lane * 4; d < a.D; d += 256) {
      const float4 q = *reinterpret_cast<const float4*>(q2 + d);
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = V4<T>::ld(ctx + (size_t)min(l0 + 4 * u, a.L - 1) * a.D + d);
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += dot4(v[u], q);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s = wave_sum(s4[u]);
      const int l = l0 + 4 * u;
      if (lane == 0 && l < a.L) { sim[l] = s; a.sim[(size_t)b * a.L + l] = s; }
    }
  }
  __syncthreads();
  const int half = a.K / 2;
  for (int i = threadIdx.x; i < 2 * a.L; i += 256) {  // conv + mask_logits
    const int which = i / a.L, l = i - which * a.L;
    const float* w = which ? a.w_ed : a.w_st;
    float v = 0.f;
    for (int k = 0; k < a.K; ++k) {
      const int j = l + k - half;
      if (j >= 0 && j < a.L) v = fmaf(w[k], sim[j], v);
    }
    const float mk = a.mask[(size_t)b * a.L + l];
    lg[i] = v * mk + (1.f - mk) * -10000.f;
  }
  __syncthreads();
  // valid-row counts of the two cross-entropies (ignore_index -1)
  float c0 = 0.f, c1 = 0.f;
  for (int i = threadIdx.x; i < a.B; i += 256) {
    c0 += a.targets[2 * i] != -1 ? 1.f : 0.f;
    c1 += a.targets[2 * i + 1] != -1 ? 1.f : 0.f;
  }
  c0 = block_sum(c0, red);
  c1 = block_sum(c1, red);
  float total = 0.f;
  for (int which = 0; which < 2; ++which) {           // log-softmax over L, wave 0 .. uniform work for all
    const float* x = lg + which * a.L;
    float mx = -3.0e38f;
    for (int l = threadIdx.x; l < a.L; l += 256) mx = fmaxf(mx, x[l]);
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int l = threadIdx.x; l < a.L; l += 256) sum += __expf(x[l] - mx);
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    float* p = (which ? a.p_ed : a.p_st) + (size_t)b * a.L;
    for (int l = threadIdx.x; l < a.L; l += 256) p[l] = __expf(x[l] - mx) * inv;
    const long long t = a.targets[2 * b + which];
    if (t != -1) total += (mx + __logf(sum) - x[t]) / (which ? c1 : c0);
    __syncthreads();
  }
  if (threadIdx.x == 0) a.loss_rows[b] = total;
}

template <typename T>
__global__ __launch_bounds__(256) void st_ed_bwd_kernel(HeroStEd a) {
  extern __shared__ float sm[];                      // dl[2][L], dsim[L], sim[L]
  __shared__ float red[4];
  float* dl = sm;
  float* dsim = sm + 2 * a.L;
  float* sim = sm + 3 * a.L;
  const int b = blockIdx.x;
  float c0 = 0.f, c1 = 0.f;
  for (int i = threadIdx.x; i < a.B; i += 256) {
    c0 += a.targets[2 * i] != -1 ? 1.f : 0.f;
    c1 += a.targets[2 * i + 1] != -1 ? 1.f : 0.f;
  }
  c0 = block_sum(c0, red);
  c1 = block_sum(c1, red);
  const float g = a.g[0] * a.g_scale;
  for (int i = threadIdx.x; i < 2 * a.L; i += 256) {  // d logits (through mask_logits)
    const int which = i / a.L, l = i - which * a.L;
    const long long t = a.targets[2 * b + which];
    float v = 0.f;
    if (t != -1) {
      const float p = (which ? a.p_ed : a.p_st)[(size_t)b * a.L + l];
      v = g * (p - (l == t ? 1.f : 0.f)) / (which ? c1 : c0) * a.mask[(size_t)b * a.L + l];
    }
    dl[i] = v;
  }
  for (int l = threadIdx.x; l < a.L; l += 256) sim[l] = a.sim[(size_t)b * a.L + l];
  __syncthreads();
  const int half = a.K / 2;
  for (int j = threadIdx.x; j < a.L; j += 256) {      // d sim[j] = sum_k w[k] * dlogit[j - k + half]
    float v = 0.f;
    for (int k = 0; k < a.K; ++k) {
      const int l = j - k + half;
      if (l >= 0 && l < a.L) v += a.w_st[k] * dl[l] + a.w_ed[k] * dl[a.L + l];
    }
    dsim[j] = v;
  }
  // conv weight gradients: dw[k] = sum_b sum_l dlogit_b[l] * sim_b[l + k - half].  Every workgroup leaves its pair's share
  // in the workspace; the LAST one to arrive (a ticket) adds the B shares up in pair order: a fixed-order sum whoever
  // comes last (round 3: every workgroup added its share with fp32 atomics, in whatever order they landed).
  if (a.dw_st || a.dw_ed) {
    float* part = a.ws + (size_t)b * (2 * (MAXK + 1));
    for (int k = 0; k < a.K; ++k) {
      float s0 = 0.f, s1 = 0.f;
      for (int l = threadIdx.x; l < a.L; l += 256) {
        const int j = l + k - half;
        if (j >= 0 && j < a.L) { s0 += dl[l] * sim[j]; s1 += dl[a.L + l] * sim[j]; }
      }
      s0 = block_sum(s0, red);
      s1 = block_sum(s1, red);
      if (threadIdx.x == 0) { part[k] = s0; part[MAXK + 1 + k] = s1; }
    }
    __shared__ int last_flag;
    __shared__ float red_t[2 * (MAXK + 1)];
    if (threadIdx.x == 0) {
      __threadfence();                                                   // release: the shares above, agent scope
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (MI355X_MICROARCH.md: never let the flag overtake the write-back)
      int* counter = reinterpret_cast<int*>(a.ws + (size_t)a.B * (2 * (MAXK + 1)));
      last_flag = atomicAdd(counter, 1) == a.B - 1;
      if (last_flag) { *counter = 0; __threadfence(); }                  // acquire (and leave the counter at zero)
    }
    __syncthreads();
    if (last_flag) {
      // Round 6: the B shares are fetched by ALL threads at once into the LDS (thread t: element t of the [B][32] table, eight
      // loads in flight), then 2 K threads add them up in pair order - the 32 serial agent-scope loads per thread of round 5
      // were most of this kernel's 24 us.  Same order of additions: the same bits.
      __shared__ float shares[64 * 2 * (MAXK + 1)];
      constexpr int W = 2 * (MAXK + 1);
      for (int b0 = 0; b0 < a.B; b0 += 64) {
        const int nb = min(64, a.B - b0);
        for (int t = threadIdx.x; t < nb * W; t += 256)
          shares[t] = __hip_atomic_load(a.ws + (size_t)b0 * W + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x < W) {
          const int k = threadIdx.x % (MAXK + 1), which = threadIdx.x / (MAXK + 1);
          if (k < a.K) {
            float t = b0 ? red_t[threadIdx.x] : 0.f;
            for (int bb = 0; bb < nb; ++bb) t += shares[bb * W + threadIdx.x];
            red_t[threadIdx.x] = t;
            if (b0 + 64 >= a.B) {
              float* dw = which ? a.dw_ed : a.dw_st;
              if (dw) dw[k] += t;
            }
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  const T* ctx = static_cast<const T*>(a.ctx) + (size_t)b * a.L * a.D;
  T* dctx = static_cast<T*>(a.dctx) + (size_t)b * a.L * a.D;
  const float* q2 = a.q2 + (size_t)b * a.D;
  for (int d = threadIdx.x * 4; d < a.D; d += 1024) {
    const float4 q = *reinterpret_cast<const float4*>(q2 + d);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int l = 0;
    for (; l + 4 <= a.L; l += 4) {                     // (round 6) four frames per trip: loads together, then the FMAs (same order) and the stores
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = V4<T>::ld(ctx + (size_t)(l + u) * a.D + d);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float s = dsim[l + u];
        acc.x = fmaf(s, v[u].x, acc.x); acc.y = fmaf(s, v[u].y, acc.y); acc.z = fmaf(s, v[u].z, acc.z); acc.w = fmaf(s, v[u].w, acc.w);
        V4<T>::st(dctx + (size_t)(l + u) * a.D + d, make_float4(s * q.x, s * q.y, s * q.z, s * q.w));
      }
    }
    for (; l < a.L; ++l) {
      const float s = dsim[l];
      const float4 v = V4<T>::ld(ctx + (size_t)l * a.D + d);
      acc.x = fmaf(s, v.x, acc.x); acc.y = fmaf(s, v.y, acc.y); acc.z = fmaf(s, v.z, acc.z); acc.w = fmaf(s, v.w, acc.w);
      V4<T>::st(dctx + (size_t)l * a.D + d, make_float4(s * q.x, s * q.y, s * q.z, s * q.w));
    }
    *reinterpret_cast<float4*>(a.dq2 + (size_t)b * a.D + d) = acc;
  }
}

}  // namespace
// out[s] = scale[s] * sum(src[s * seg_len .. (s + 1) * seg_len)) for up to 4 segments: the final reductions of the loss head
// (sum of the start / end rows, means of the two ranking-loss rows) with their loss weights folded in - one wave per segment,
// fixed summation order (lane-strided partial sums, then the DPP wave sum): bit-reproducible.
struct SumsArgs { const float* src; float* out; int n_segs, seg_len; float scale[4]; };
__global__ __launch_bounds__(64) void sums_scaled_kernel(SumsArgs a) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const float* p = a.src + (size_t)s * a.seg_len;
  float v = 0.f;
  for (int i = lane; i < a.seg_len; i += 64) v += p[i];
  v = wave_sum(v);
  if (lane == 0) a.out[s] = v * a.scale[s];
}

}  // namespace hero

using namespace hero;

#define HERO_BY_DTYPE(dt, KERNEL, grid, lds, s, arg, what)                                         \
  do {                                                                                             \
    if ((dt) == HERO_F32) hipLaunchKernelGGL((KERNEL<float>), dim3(grid), dim3(256), lds, s, arg);  \
    else if ((dt) == HERO_BF16) hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(grid), dim3(256), lds, s, arg); \
    else { set_error(what ": bad dtype %d", (dt)); return HERO_ERR_ARG; }                           \
    return check_launch(what);                                                                     \
  } while (0)

extern "C" int hero_query_pool_fwd(const HeroQueryPool* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->q && a->mask && a->w && a->pooled && a->att, "hero_query_pool_fwd: null pointer");
  HERO_REQUIRE(a->L > 0 && a->L <= 4096 && a->D > 0 && a->D % 4 == 0, "hero_query_pool_fwd: bad dims L=%d D=%d", a->L, a->D);
  if (a->B <= 0) return HERO_OK;
  HERO_BY_DTYPE(a->dtype, query_pool_fwd_kernel, a->B, a->L * sizeof(float), static_cast<hipStream_t>(stream), *a, "hero_query_pool_fwd");
}
extern "C" int hero_query_pool_bwd(const HeroQueryPool* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->q && a->mask && a->w && a->att && a->dpooled && a->dq, "hero_query_pool_bwd: null pointer");
  HERO_REQUIRE(a->L > 0 && a->L <= 4096 && a->D > 0 && a->D % 4 == 0, "hero_query_pool_bwd: bad dims L=%d D=%d", a->L, a->D);
  if (a->B <= 0) return HERO_OK;
  HERO_BY_DTYPE(a->dtype, query_pool_bwd_kernel, a->B, 2 * a->L * sizeof(float), static_cast<hipStream_t>(stream), *a, "hero_query_pool_bwd");
}

extern "C" int hero_rownorm_fwd(const HeroRowNorm* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->x && a->y && a->rnorm, "hero_rownorm_fwd: null pointer");
  HERO_REQUIRE(a->cols > 0 && a->cols % 4 == 0, "hero_rownorm_fwd: cols (%d) must be a positive multiple of 4", a->cols);
  if (a->rows <= 0) return HERO_OK;
  HERO_BY_DTYPE(a->x_dtype, rownorm_fwd_kernel, (a->rows + 3) / 4, 0, static_cast<hipStream_t>(stream), *a, "hero_rownorm_fwd");
}
extern "C" int hero_rownorm_bwd(const HeroRowNorm* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->x && a->rnorm && a->dy && a->dx, "hero_rownorm_bwd: null pointer");
  HERO_REQUIRE(a->cols > 0 && a->cols % 4 == 0, "hero_rownorm_bwd: cols (%d) must be a positive multiple of 4", a->cols);
  if (a->rows <= 0) return HERO_OK;
  HERO_BY_DTYPE(a->x_dtype, rownorm_bwd_kernel, (a->rows + 3) / 4, 0, static_cast<hipStream_t>(stream), *a, "hero_rownorm_bwd");
}

extern "C" int hero_score_max_fwd(const HeroScoreMax* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->s && a->mask && a->out && a->arg, "hero_score_max_fwd: null pointer");
  HERO_REQUIRE(a->L > 0 && a->ld_s >= a->N * a->L, "hero_score_max_fwd: bad L=%d / ld_s=%d", a->L, a->ld_s);
  if (a->M <= 0 || a->N <= 0) return HERO_OK;
  hipLaunchKernelGGL(score_max_fwd_kernel, dim3((a->M * a->N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("hero_score_max_fwd");
}
extern "C" int hero_score_max_bwd(const HeroScoreMax* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->mask && a->arg && a->ds_ctx && a->ds_q && a->gc && a->gq && a->qn && a->cn && a->dqn && a->dcn,
               "hero_score_max_bwd: null pointer");
  HERO_REQUIRE(a->L > 0 && a->D > 0 && a->D % 4 == 0, "hero_score_max_bwd: bad dims L=%d D=%d", a->L, a->D);
  HERO_REQUIRE(a->n0 >= 0 && a->n_own >= 0 && a->n0 + a->n_own <= a->N, "hero_score_max_bwd: own range [%d, +%d) outside N=%d", a->n0,
               a->n_own, a->N);
  if (a->M <= 0 || a->N <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  HERO_REQUIRE(a->N <= 4096, "hero_score_max_bwd: N = %d beyond the 4096 (weight, frame) pairs staged in the LDS", a->N);
  hipLaunchKernelGGL(score_max_bwd_q_kernel, dim3(a->M), dim3(256), (size_t)a->N * 8, s, *a);
  int rc = check_launch("hero_score_max_bwd(q)");
  if (rc || a->n_own == 0) return rc;
  hipLaunchKernelGGL(score_max_bwd_c_kernel, dim3(a->n_own), dim3(256), 0, s, *a);
  return check_launch("hero_score_max_bwd(c)");
}

extern "C" int hero_rank_loss(const HeroRankLoss* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->s && a->loss_ctx_rows && a->loss_q_rows && a->ds_ctx && a->ds_q, "hero_rank_loss: null pointer");
  HERO_REQUIRE(a->nv > 1 && a->nq >= a->nv && a->nq % a->nv == 0, "hero_rank_loss: need nv > 1 and nq a multiple of nv (nq=%d nv=%d)",
               a->nq, a->nv);
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    const size_t n = (size_t)a->nq * a->nv;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, s, a->ds_q, n);
  }
  hipLaunchKernelGGL(rank_loss_kernel, dim3(a->nq, 2), dim3(256), 0, s, *a);
  return check_launch("hero_rank_loss");
}

extern "C" int hero_st_ed_fwd(const HeroStEd* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->q2 && a->ctx && a->mask && a->w_st && a->w_ed && a->targets && a->loss_rows && a->p_st && a->p_ed && a->sim,
               "hero_st_ed_fwd: null pointer");
  HERO_REQUIRE(a->L > 0 && a->L <= 2048 && a->D % 4 == 0 && a->K >= 1 && a->K <= MAXK && (a->K & 1), "hero_st_ed_fwd: bad dims L=%d D=%d K=%d",
               a->L, a->D, a->K);
  if (a->B <= 0) return HERO_OK;
  HERO_BY_DTYPE(a->dtype, st_ed_fwd_kernel, a->B, 3 * a->L * sizeof(float), static_cast<hipStream_t>(stream), *a, "hero_st_ed_fwd");
}
extern "C" size_t hero_st_ed_bwd_workspace_bytes(int B) { return ((size_t)(B > 0 ? B : 0) * 2 * (MAXK + 1) + 4) * sizeof(float); }

extern "C" int hero_st_ed_bwd(const HeroStEd* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->q2 && a->ctx && a->mask && a->w_st && a->w_ed && a->targets && a->p_st && a->p_ed && a->sim && a->g && a->dq2 && a->dctx,
               "hero_st_ed_bwd: null pointer");
  HERO_REQUIRE(a->L > 0 && a->L <= 2048 && a->D % 4 == 0 && a->K >= 1 && a->K <= MAXK && (a->K & 1), "hero_st_ed_bwd: bad dims L=%d D=%d K=%d",
               a->L, a->D, a->K);
  HERO_REQUIRE(a->ws || !(a->dw_st || a->dw_ed), "hero_st_ed_bwd: dw_st / dw_ed need the workspace (hero_st_ed_bwd_workspace_bytes)");
  if (a->B <= 0) return HERO_OK;
  HERO_BY_DTYPE(a->dtype, st_ed_bwd_kernel, a->B, 4 * a->L * sizeof(float), static_cast<hipStream_t>(stream), *a, "hero_st_ed_bwd");
}

extern "C" int hero_sums_scaled(const float* src, int n_segs, int seg_len, const float* scales, float* out, hero_stream_t stream) {
  HERO_REQUIRE(src && out && scales && n_segs >= 1 && n_segs <= 4 && seg_len >= 1, "hero_sums_scaled: 1..4 segments of >= 1 element");
  SumsArgs a;
  a.src = src; a.out = out; a.n_segs = n_segs; a.seg_len = seg_len;
  for (int i = 0; i < 4; ++i) a.scale[i] = i < n_segs ? scales[i] : 0.f;
  hipLaunchKernelGGL(sums_scaled_kernel, dim3(n_segs), dim3(64), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("hero_sums_scaled");
}
