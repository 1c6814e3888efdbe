// LayerNorm forward / backward (HBM-bound) for gfx950.
//
// forward : one 64-lane wave per row, the whole row lives in registers (VPL float4 per lane),
//           exact two-pass statistics in fp32, optional gathered embedding-table rows added in
//           front (SubEmbeddings / ImageEmbeddings / FrameEmbeddings sums) and dropout behind.
// backward: dx with the same row-in-registers scheme; dgamma/dbeta and plain bias gradients by a
//           two-stage deterministic column reduction (partials in a caller workspace, no atomics).
#include "common.h"

namespace hero {

// General forward (any cols % 4 == 0, optional x, up to three gathered table rows).  Written in PHASES so that the
// loads of a phase are in flight together: per-chunk branches (`if (c < cols)`, `if (x)`, `if (t[k])`) make every
// chunk a basic block and the compiler waits vmcnt(0) right after each load (68 serial round trips for a 4352-wide
// row).  A chunk past the row reads a clamped address and is multiplied by 0.
template <typename TX, typename TY, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(HeroLnFwd a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.rows) return;
  const int cols = a.cols;
  int cc[VPL];
  float okf[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    cc[i] = min(c, cols - 4);
    okf[i] = c < cols ? 1.f : 0.f;
  }
  float4 v[VPL];
  if (a.x) {                                          // uniform
    const TX* x = static_cast<const TX*>(a.x) + (size_t)row * cols;
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = V4<TX>::ld(x + cc[i]);
  } else {
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (a.tab[k]) {                                   // uniform
      const float* t = a.tab[k] + (size_t)(a.idx[k] ? a.idx[k][row] : 0) * cols;
      float4 w[VPL];
#pragma unroll
      for (int i = 0; i < VPL; ++i) w[i] = *reinterpret_cast<const float4*>(t + cc[i]);
#pragma unroll
      for (int i = 0; i < VPL; ++i) { v[i].x += w[i].x; v[i].y += w[i].y; v[i].z += w[i].z; v[i].w += w[i].w; }
    }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i].x *= okf[i]; v[i].y *= okf[i]; v[i].z *= okf[i]; v[i].w *= okf[i];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
    q += ((dx * dx + dy * dy) + (dz * dz + dw * dw)) * okf[i];
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + a.eps);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
  DropCtx drop(a.dropout);
  TY* y = static_cast<TY*>(a.y) + (size_t)row * cols;
  TY* pre = a.pre ? static_cast<TY*>(a.pre) + (size_t)row * cols : nullptr;
  constexpr int GRP = 4;                              // gamma / beta of GRP chunks are fetched together, then GRP stores
#pragma unroll
  for (int i0 = 0; i0 < VPL; i0 += GRP) {
    float4 g[GRP], b[GRP];
#pragma unroll
    for (int u = 0; u < GRP; ++u)
      if (i0 + u < VPL) {
        g[u] = *reinterpret_cast<const float4*>(a.gamma + cc[i0 + u]);
        b[u] = *reinterpret_cast<const float4*>(a.beta + cc[i0 + u]);
      }
#pragma unroll
    for (int u = 0; u < GRP; ++u)
      if (i0 + u < VPL) {
        const int i = i0 + u;
        float4 o;
        o.x = (v[i].x - mean) * rstd * g[u].x + b[u].x;
        o.y = (v[i].y - mean) * rstd * g[u].y + b[u].y;
        o.z = (v[i].z - mean) * rstd * g[u].z + b[u].z;
        o.w = (v[i].w - mean) * rstd * g[u].w + b[u].w;
        if (drop.on()) {
          const float4 m = drop.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)cc[i]) >> 2);
          o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
        }
        if (okf[i] != 0.f) {
          if (pre) V4<TY>::st(pre + cc[i], v[i]);
          V4<TY>::st(y + cc[i], o);
        }
      }
  }
}

// Straight-line form for the common case (x given, no embedding tables, cols == VPL * 256): no per-chunk branches, so all
// loads of a row - and gamma / beta - are in flight together.  In the general kernel above every chunk sits in its
// own basic block (`if (c < cols)`, `if (x)`, `if (t[k])`) and the compiler waits vmcnt(0) right after each load:
// three serial round trips per 768-wide row.
template <typename TX, typename TY, int VPL>
__global__ __launch_bounds__(256) void ln_fwd_full_kernel(HeroLnFwd a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.rows) return;
  constexpr int cols = VPL * 256;
  const TX* x = static_cast<const TX*>(a.x) + (size_t)row * cols;
  float4 v[VPL], g[VPL], b[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) v[i] = V4<TX>::ld(x + (lane + 64 * i) * 4);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    g[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
    b[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + a.eps);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
  DropCtx drop(a.dropout);
  TY* y = static_cast<TY*>(a.y) + (size_t)row * cols;
  TY* pre = a.pre ? static_cast<TY*>(a.pre) + (size_t)row * cols : nullptr;
  float4 o[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    o[i].x = (v[i].x - mean) * rstd * g[i].x + b[i].x;
    o[i].y = (v[i].y - mean) * rstd * g[i].y + b[i].y;
    o[i].z = (v[i].z - mean) * rstd * g[i].z + b[i].z;
    o[i].w = (v[i].w - mean) * rstd * g[i].w + b[i].w;
  }
  if (drop.on()) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float4 m = drop.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)((lane + 64 * i) * 4)) >> 2);
      o[i].x *= m.x; o[i].y *= m.y; o[i].z *= m.z; o[i].w *= m.w;
    }
  }
  if (pre) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) V4<TY>::st(pre + (lane + 64 * i) * 4, v[i]);
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) V4<TY>::st(y + (lane + 64 * i) * 4, o[i]);
}

template <typename TX, typename T, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(HeroLnBwd a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.rows) return;
  const int cols = a.cols;
  const TX* x = static_cast<const TX*>(a.x) + (size_t)row * cols;
  const T* dy = static_cast<const T*>(a.dy) + (size_t)row * cols;
  const float mean = a.mean[row], rstd = a.rstd[row];
  DropCtx dout(a.dropout_out);
  float4 xh[VPL], g[VPL];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    g[i] = xh[i];
    if (c < cols) {
      const float4 xv = V4<TX>::ld(x + c);
      float4 d = V4<T>::ld(dy + c);
      if (dout.on()) {
        const float4 m = dout.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)c) >> 2);
        d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
      }
      const float4 gm = *reinterpret_cast<const float4*>(a.gamma + c);
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      g[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
    }
  }
  const float inv = 1.f / (float)cols;
  s1 = wave_sum(s1) * inv;
  s2 = wave_sum(s2) * inv;
  DropCtx din(a.dropout_in);
  T* dx = a.dx ? static_cast<T*>(a.dx) + (size_t)row * cols : nullptr;
  T* dxd = a.dx_dropped ? static_cast<T*>(a.dx_dropped) + (size_t)row * cols : nullptr;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < cols) {
      float4 o;
      o.x = rstd * (g[i].x - s1 - xh[i].x * s2);
      o.y = rstd * (g[i].y - s1 - xh[i].y * s2);
      o.z = rstd * (g[i].z - s1 - xh[i].z * s2);
      o.w = rstd * (g[i].w - s1 - xh[i].w * s2);
      if (dx) V4<T>::st(dx + c, o);
      if (dxd) {
        if (din.on()) {
          const float4 m = din.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)c) >> 2);
          o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
        }
        V4<T>::st(dxd + c, o);
      }
    }
  }
}

// Fused backward for rows <= 1024 wide: dx (and dx * mask_in) as above, PLUS per-workgroup partial
// sums of dgamma = sum dy*xhat, dbeta = sum dy and dbias_in = sum dx*mask_in (the bias gradient of
// the linear layer that feeds this LayerNorm) in the same pass over x and dy.  Each wave walks rows
// grid-stride and keeps the column partials in registers; one LDS reduction per workgroup writes
// partial[block][3][cols]; colred_final3_kernel folds the partials (deterministic, no atomics).
template <typename TX, typename T, int VPL>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(HeroLnBwd a, float* partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];       // [4 waves][3][cols]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cols = a.cols;
  DropCtx dout(a.dropout_out), din(a.dropout_in);
  float4 ag[VPL], ab[VPL], ai[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) ag[i] = ab[i] = ai[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float inv = 1.f / (float)cols;
  for (int row = blockIdx.x * 4 + wave; row < a.rows; row += gridDim.x * 4) {
    const TX* x = static_cast<const TX*>(a.x) + (size_t)row * cols;
    const T* dy = static_cast<const T*>(a.dy) + (size_t)row * cols;
    const float mean = a.mean[row], rstd = a.rstd[row];
    float4 xh[VPL], g[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 4;
      xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      g[i] = xh[i];
      if (c < cols) {
        const float4 xv = V4<TX>::ld(x + c);
        float4 d = V4<T>::ld(dy + c);
        if (dout.on()) {
          const float4 m = dout.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)c) >> 2);
          d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
        }
        const float4 gm = *reinterpret_cast<const float4*>(a.gamma + c);
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        g[i] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
        ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      }
    }
    s1 = wave_sum(s1) * inv;
    s2 = wave_sum(s2) * inv;
    T* dx = a.dx ? static_cast<T*>(a.dx) + (size_t)row * cols : nullptr;
    T* dxd = a.dx_dropped ? static_cast<T*>(a.dx_dropped) + (size_t)row * cols : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < cols) {
        float4 o;
        o.x = rstd * (g[i].x - s1 - xh[i].x * s2);
        o.y = rstd * (g[i].y - s1 - xh[i].y * s2);
        o.z = rstd * (g[i].z - s1 - xh[i].z * s2);
        o.w = rstd * (g[i].w - s1 - xh[i].w * s2);
        if (dx) V4<T>::st(dx + c, o);
        if (din.on()) {
          const float4 m = din.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)c) >> 2);
          o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
        }
        if (dxd) V4<T>::st(dxd + c, o);
        ai[i].x += o.x; ai[i].y += o.y; ai[i].z += o.z; ai[i].w += o.w;
      }
    }
  }
  // ---- workgroup reduction of the three column partials
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < cols) {
      *reinterpret_cast<float4*>(red + ((size_t)(wave * 3 + 0)) * cols + c) = ag[i];
      *reinterpret_cast<float4*>(red + ((size_t)(wave * 3 + 1)) * cols + c) = ab[i];
      *reinterpret_cast<float4*>(red + ((size_t)(wave * 3 + 2)) * cols + c) = ai[i];
    }
  }
  __syncthreads();
  const int n4 = 3 * (cols >> 2);
  for (int q = threadIdx.x; q < n4; q += 256) {
    const int k = q / (cols >> 2), c = (q - k * (cols >> 2)) * 4;
    float4 v = *reinterpret_cast<const float4*>(red + (size_t)k * cols + c);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(red + ((size_t)(w * 3 + k)) * cols + c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + k) * cols + c) = v;
  }
}

// Straight-line form of the fused backward (cols == VPL * 256): gamma lives in registers for the whole kernel and
// all loads of a row are issued before anything is used (the general kernel waits vmcnt(0) after each chunk's load
// because every chunk is a basic block of its own).
template <typename TX, typename T, int VPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void ln_bwd_fused_full_kernel(HeroLnBwd a, float* partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];       // [4 waves][3][cols]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int cols = VPL * 256;
  DropCtx dout(a.dropout_out), din(a.dropout_in);
  float4 ag[VPL], ab[VPL], ai[VPL], gm[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    ag[i] = ab[i] = ai[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    gm[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
  }
  const float inv = 1.f / (float)cols;
  for (int row = blockIdx.x * 4 + wave; row < a.rows; row += gridDim.x * 4) {
    const TX* x = static_cast<const TX*>(a.x) + (size_t)row * cols;
    const T* dy = static_cast<const T*>(a.dy) + (size_t)row * cols;
    float4 xv[VPL], d[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      xv[i] = V4<TX>::ld(x + (lane + 64 * i) * 4);
      d[i] = V4<T>::ld(dy + (lane + 64 * i) * 4);
    }
    const float mean = a.mean[row], rstd = a.rstd[row];
    if (dout.on()) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const float4 m = dout.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)((lane + 64 * i) * 4)) >> 2);
        d[i].x *= m.x; d[i].y *= m.y; d[i].z *= m.z; d[i].w *= m.w;
      }
    }
    float4 xh[VPL], g[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      xh[i] = make_float4((xv[i].x - mean) * rstd, (xv[i].y - mean) * rstd, (xv[i].z - mean) * rstd, (xv[i].w - mean) * rstd);
      g[i] = make_float4(d[i].x * gm[i].x, d[i].y * gm[i].y, d[i].z * gm[i].z, d[i].w * gm[i].w);
      ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
      ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
    }
    s1 = wave_sum(s1) * inv;
    s2 = wave_sum(s2) * inv;
    T* dx = a.dx ? static_cast<T*>(a.dx) + (size_t)row * cols : nullptr;
    T* dxd = a.dx_dropped ? static_cast<T*>(a.dx_dropped) + (size_t)row * cols : nullptr;
    float4 o[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      o[i].x = rstd * (g[i].x - s1 - xh[i].x * s2);
      o[i].y = rstd * (g[i].y - s1 - xh[i].y * s2);
      o[i].z = rstd * (g[i].z - s1 - xh[i].z * s2);
      o[i].w = rstd * (g[i].w - s1 - xh[i].w * s2);
    }
    if (dx) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) V4<T>::st(dx + (lane + 64 * i) * 4, o[i]);
    }
    if (din.on()) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const float4 m = din.mask4(((uint64_t)row * (uint64_t)cols + (uint64_t)((lane + 64 * i) * 4)) >> 2);
        o[i].x *= m.x; o[i].y *= m.y; o[i].z *= m.z; o[i].w *= m.w;
      }
    }
    if (dxd) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) V4<T>::st(dxd + (lane + 64 * i) * 4, o[i]);
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) { ai[i].x += o[i].x; ai[i].y += o[i].y; ai[i].z += o[i].z; ai[i].w += o[i].w; }
  }
  // ---- workgroup reduction of the three column partials (as in ln_bwd_fused_kernel)
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (lane + 64 * i) * 4;
    *reinterpret_cast<float4*>(red + ((size_t)(wave * 3 + 0)) * cols + c) = ag[i];
    *reinterpret_cast<float4*>(red + ((size_t)(wave * 3 + 1)) * cols + c) = ab[i];
    *reinterpret_cast<float4*>(red + ((size_t)(wave * 3 + 2)) * cols + c) = ai[i];
  }
  __syncthreads();
  constexpr int n4 = 3 * (cols >> 2);
  for (int q = threadIdx.x; q < n4; q += 256) {
    const int k = q / (cols >> 2), c = (q - k * (cols >> 2)) * 4;
    float4 v = *reinterpret_cast<const float4*>(red + (size_t)k * cols + c);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(red + ((size_t)(w * 3 + k)) * cols + c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + k) * cols + c) = v;
  }
}

// out_k[c] = beta*out_k[c] + sum_b partial[b][k][c] for k = blockIdx.y (outputs may be NULL).
// 16 float4 column groups x 16 partial-lanes per workgroup; fixed summation order.
// gridDim.z > 1 (only with beta == 1): each z-slice folds its share of the partials and adds it with
// fp32 atomics (8-way contention) - 8x the parallelism of the single-slice, deterministic form.
__global__ __launch_bounds__(256) void colred_final3_kernel(const float* partial, float* o0, float* o1, float* o2, int cols,
                                                            int nblocks, float beta) {
  const int k = blockIdx.y;
  float* out = k == 0 ? o0 : (k == 1 ? o1 : o2);
  if (!out) return;
  __shared__ float4 red[16][16];
  const int cg = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + cg) * 4;
  const int per = (nblocks + gridDim.z - 1) / gridDim.z;
  const int b0 = blockIdx.z * per, b1 = min(nblocks, b0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols) {
    int b = b0 + kl;
    // four partials per trip, loads issued together (one per trip = one 16-byte request in flight per thread: the
    // 8 trips of the 1024-block fold were 8 serial round trips, most of the kernel's 5 us)
    for (; b + 48 < b1; b += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(partial + ((size_t)(b + 16 * u) * 3 + k) * cols + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; b < b1; b += 16) {
      const float4 v = *reinterpret_cast<const float4*>(partial + ((size_t)b * 3 + k) * cols + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[kl][cg] = s;
  __syncthreads();
  if (kl == 0 && c < cols) {
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      const float4 v = red[j][cg];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (gridDim.z > 1) {
      atomicAdd(out + c, s.x); atomicAdd(out + c + 1, s.y); atomicAdd(out + c + 2, s.z); atomicAdd(out + c + 3, s.w);
      return;
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (beta != 0.f) {
      o = *reinterpret_cast<const float4*>(out + c);
      o.x *= beta; o.y *= beta; o.z *= beta; o.w *= beta;
    }
    o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
    *reinterpret_cast<float4*>(out + c) = o;
  }
}

// partial column sums over a chunk of rows.  pb[chunk][c] = sum dy_eff ; pg[chunk][c] = sum dy_eff*xhat
struct ColRed {
  float* og;          // atomic mode: accumulate straight into the outputs (beta == 1), no second pass
  float* ob;
  int atomic;
  const void* x;      // [rows, cols] TX or null
  const void* dy;     // [rows, ld]  T
  const float* mean;
  const float* rstd;
  float* pg;
  float* pb;
  int rows, cols, ld, rows_per_chunk;
  HeroDropout dropout;
};

template <typename TX, typename T>
__global__ __launch_bounds__(256) void colred_kernel(ColRed a) {
  __shared__ float4 red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + tx) * 4;
  const int chunk = blockIdx.y;
  const int r0 = chunk * a.rows_per_chunk;
  const int r1 = min(a.rows, r0 + a.rows_per_chunk);
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
  if (c < a.cols) {
    DropCtx drop(a.dropout);
    const TX* x = static_cast<const TX*>(a.x);
    const T* dy = static_cast<const T*>(a.dy);
    int r = r0 + ty;
    // four rows per trip, their loads issued together (one load per trip left a single 8-byte request in flight per
    // wave: 12 serial round trips per chunk)
    for (; r + 12 < r1; r += 16) {
      float4 d[4], xv[4];
      float mu[4], rs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) d[u] = V4<T>::ld(dy + (size_t)(r + 4 * u) * a.ld + c);
      if (x) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = V4<TX>::ld(x + (size_t)(r + 4 * u) * a.cols + c);
          mu[u] = a.mean[r + 4 * u];
          rs[u] = a.rstd[r + 4 * u];
        }
      }
      if (drop.on()) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 m = drop.mask4(((uint64_t)(r + 4 * u) * (uint64_t)a.cols + (uint64_t)c) >> 2);
          d[u].x *= m.x; d[u].y *= m.y; d[u].z *= m.z; d[u].w *= m.w;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { sb.x += d[u].x; sb.y += d[u].y; sb.z += d[u].z; sb.w += d[u].w; }
      if (x) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          sg.x += d[u].x * (xv[u].x - mu[u]) * rs[u]; sg.y += d[u].y * (xv[u].y - mu[u]) * rs[u];
          sg.z += d[u].z * (xv[u].z - mu[u]) * rs[u]; sg.w += d[u].w * (xv[u].w - mu[u]) * rs[u];
        }
      }
    }
    for (; r < r1; r += 4) {
      float4 d = V4<T>::ld(dy + (size_t)r * a.ld + c);
      if (drop.on()) {
        const float4 m = drop.mask4(((uint64_t)r * (uint64_t)a.cols + (uint64_t)c) >> 2);
        d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
      }
      sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
      if (x) {
        const float4 xv = V4<TX>::ld(x + (size_t)r * a.cols + c);
        const float mu = a.mean[r], rs = a.rstd[r];
        sg.x += d.x * (xv.x - mu) * rs; sg.y += d.y * (xv.y - mu) * rs;
        sg.z += d.z * (xv.z - mu) * rs; sg.w += d.w * (xv.w - mu) * rs;
      }
    }
  }
  red[0][ty][tx] = sg;
  red[1][ty][tx] = sb;
  __syncthreads();
  if (ty == 0 && c < a.cols) {
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 g2 = red[0][k][tx], b2 = red[1][k][tx];
      sg.x += g2.x; sg.y += g2.y; sg.z += g2.z; sg.w += g2.w;
      sb.x += b2.x; sb.y += b2.y; sb.z += b2.z; sb.w += b2.w;
    }
    if (a.atomic) {
      if (a.og) { atomicAdd(a.og + c, sg.x); atomicAdd(a.og + c + 1, sg.y); atomicAdd(a.og + c + 2, sg.z); atomicAdd(a.og + c + 3, sg.w); }
      if (a.ob) { atomicAdd(a.ob + c, sb.x); atomicAdd(a.ob + c + 1, sb.y); atomicAdd(a.ob + c + 2, sb.z); atomicAdd(a.ob + c + 3, sb.w); }
    } else {
      if (a.pg) *reinterpret_cast<float4*>(a.pg + (size_t)chunk * a.cols + c) = sg;
      *reinterpret_cast<float4*>(a.pb + (size_t)chunk * a.cols + c) = sb;
    }
  }
}

// out[c] = beta*out[c] + sum_k partial[k][c]: 16 columns x 16 chunk-lanes per block, lanes of one
// column combine with shuffles (fixed order -> deterministic).
__global__ __launch_bounds__(256) void colred_final_kernel(const float* pg, const float* pb, float* og, float* ob, int cols,
                                                           int nchunks, float beta) {
  const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  // gridDim.z > 1 (only with beta == 1): each z-slice folds its share of the chunks and adds it with
  // fp32 atomics (8-way contention), as in colred_final3_kernel
  const int per = (nchunks + gridDim.z - 1) / gridDim.z;
  const int k0 = blockIdx.z * per, k1 = min(nchunks, k0 + per);
  float sg = 0.f, sb = 0.f;
  if (c < cols) {
    int k = k0 + kl;
    for (; k + 48 < k1; k += 64) {                     // four chunks per trip, loads issued together
      float g4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
      if (og) {
#pragma unroll
        for (int u = 0; u < 4; ++u) g4[u] = pg[(size_t)(k + 16 * u) * cols + c];
      }
      if (ob) {
#pragma unroll
        for (int u = 0; u < 4; ++u) b4[u] = pb[(size_t)(k + 16 * u) * cols + c];
      }
      sg += (g4[0] + g4[1]) + (g4[2] + g4[3]);
      sb += (b4[0] + b4[1]) + (b4[2] + b4[3]);
    }
    for (; k < k1; k += 16) {
      if (og) sg += pg[(size_t)k * cols + c];
      if (ob) sb += pb[(size_t)k * cols + c];
    }
  }
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    sg += __shfl_xor(sg, o, 64);
    sb += __shfl_xor(sb, o, 64);
  }
  __shared__ float red[2][4][16];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 16) { red[0][wave][cl] = sg; red[1][wave][cl] = sb; }
  __syncthreads();
  if (threadIdx.x < 16 && c < cols) {
    sg = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    sb = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    if (gridDim.z > 1) {
      if (og) atomicAdd(og + c, sg);
      if (ob) atomicAdd(ob + c, sb);
      return;
    }
    if (og) og[c] = (beta != 0.f ? beta * og[c] : 0.f) + sg;
    if (ob) ob[c] = (beta != 0.f ? beta * ob[c] : 0.f) + sb;
  }
}

static inline int chunking(int rows, int* rpc) {
  int r = (rows + 255) / 256;
  if (r < 16) r = 16;
  *rpc = r;
  return (rows + r - 1) / r;
}

template <typename TX, typename T>
static int run_colred(const void* x, const void* dy, const float* mean, const float* rstd, float* og, float* ob, int rows,
                      int cols, int ld, float beta, const HeroDropout& dr, void* ws, hipStream_t s) {
  ColRed a;
  a.x = x; a.dy = dy; a.mean = mean; a.rstd = rstd;
  a.rows = rows; a.cols = cols; a.ld = ld; a.dropout = dr;
  a.og = x ? og : nullptr;
  a.ob = ob;
  const int nchunks = chunking(rows, &a.rows_per_chunk);
  a.pb = static_cast<float*>(ws);
  a.pg = x ? a.pb + (size_t)nchunks * cols : nullptr;
  a.atomic = 0;   // (fp32 atomics straight into the outputs were measured SLOWER, at 256-way and at 64-way contention)
  hipLaunchKernelGGL((colred_kernel<TX, T>), dim3((cols + 255) / 256, nchunks), dim3(256), 0, s, a);
  int rc = check_launch("colred");
  if (rc || a.atomic) return rc;
  // (round 3 cut long folds into 8 z-slices that added their shares with fp32 atomics: order-dependent sums; one slice
  // folds 256 partial rows in a few microseconds, and the result is the same every run)
  const int zs = 1;
  (void)beta;
  hipLaunchKernelGGL(colred_final_kernel, dim3((cols + 15) / 16, 1, zs), dim3(256), 0, s, a.pg, a.pb, x ? og : nullptr, ob, cols,
                     nchunks, beta);
  return check_launch("colred_final");
}

#define HERO_VPL_SWITCH(cols, CALL)                                        \
  do {                                                                     \
    const int need = ((cols) + 255) / 256;                                 \
    if (need <= 1) { CALL(1); }                                            \
    else if (need <= 2) { CALL(2); }                                       \
    else if (need <= 3) { CALL(3); }                                       \
    else if (need <= 4) { CALL(4); }                                       \
    else if (need <= 6) { CALL(6); }                                       \
    else if (need <= 8) { CALL(8); }                                       \
    else if (need <= 12) { CALL(12); }                                     \
    else if (need <= 17) { CALL(17); }                                     \
    else if (need <= 24) { CALL(24); }                                     \
    else { set_error("layernorm: cols %d too wide (max 6144)", (cols)); return HERO_ERR_UNSUPPORTED; } \
  } while (0)

}  // namespace hero

#define LN_BWD_MAX_BLOCKS 1024
using namespace hero;

extern "C" int hero_layernorm_fwd(const HeroLnFwd* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->y && a->gamma && a->beta, "hero_layernorm_fwd: null pointer");
  HERO_REQUIRE(a->x || a->tab[0] || a->tab[1] || a->tab[2], "hero_layernorm_fwd: no input");
  HERO_REQUIRE(a->cols > 0 && a->cols % 4 == 0, "hero_layernorm_fwd: cols (%d) must be a positive multiple of 4", a->cols);
  if (a->rows <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((a->rows + 3) / 4), block(256);
  const int xd = a->x ? a->x_dtype : a->y_dtype, yd = a->y_dtype;
  if (a->x && !a->tab[0] && !a->tab[1] && !a->tab[2] && a->cols % 256 == 0 && a->cols <= 1024) {   // straight-line kernel
#define CALLF(V)                                                                                                           \
  if (xd == HERO_F32 && yd == HERO_F32) hipLaunchKernelGGL((ln_fwd_full_kernel<float, float, V>), grid, block, 0, s, *a);        \
  else if (xd == HERO_F32 && yd == HERO_BF16) hipLaunchKernelGGL((ln_fwd_full_kernel<float, bf16_t, V>), grid, block, 0, s, *a); \
  else if (xd == HERO_BF16 && yd == HERO_BF16) hipLaunchKernelGGL((ln_fwd_full_kernel<bf16_t, bf16_t, V>), grid, block, 0, s, *a); \
  else { set_error("hero_layernorm_fwd: unsupported dtypes x=%d y=%d", xd, yd); return HERO_ERR_UNSUPPORTED; }
    const int v = a->cols / 256;
    if (v == 1) { CALLF(1); } else if (v == 2) { CALLF(2); } else if (v == 3) { CALLF(3); } else { CALLF(4); }
#undef CALLF
    return check_launch("hero_layernorm_fwd(full)");
  }
#define CALL(V)                                                                                                   \
  if (xd == HERO_F32 && yd == HERO_F32) hipLaunchKernelGGL((ln_fwd_kernel<float, float, V>), grid, block, 0, s, *a);        \
  else if (xd == HERO_F32 && yd == HERO_BF16) hipLaunchKernelGGL((ln_fwd_kernel<float, bf16_t, V>), grid, block, 0, s, *a); \
  else if (xd == HERO_BF16 && yd == HERO_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, bf16_t, V>), grid, block, 0, s, *a); \
  else { set_error("hero_layernorm_fwd: unsupported dtypes x=%d y=%d", xd, yd); return HERO_ERR_UNSUPPORTED; }
  HERO_VPL_SWITCH(a->cols, CALL);
#undef CALL
  return check_launch("hero_layernorm_fwd");
}

extern "C" size_t hero_layernorm_bwd_workspace_bytes(int rows, int cols) {
  (void)rows;
  return (size_t)LN_BWD_MAX_BLOCKS * (size_t)cols * 3 * sizeof(float);
}
extern "C" size_t hero_colsum_workspace_bytes(int rows, int cols) {
  (void)rows;
  return (size_t)256 * (size_t)cols * sizeof(float);
}

extern "C" int hero_layernorm_bwd(const HeroLnBwd* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->x && a->dy && a->gamma && a->mean && a->rstd, "hero_layernorm_bwd: null pointer");
  HERO_REQUIRE(a->cols > 0 && a->cols % 4 == 0, "hero_layernorm_bwd: cols (%d) must be a positive multiple of 4", a->cols);
  if (a->rows <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int xd = a->x_dtype, d = a->dtype;
  const bool want_params = a->dgamma || a->dbeta || a->dbias_in;
  if (want_params) HERO_REQUIRE(a->workspace, "hero_layernorm_bwd: workspace required for parameter gradients");
  // ---- fused single pass (rows up to 1024 wide)
  if (want_params && (a->dx || a->dx_dropped || a->dbias_in) && a->cols <= 1024) {
    int nblk = (a->rows + 3) / 4;
    if (nblk > LN_BWD_MAX_BLOCKS) nblk = LN_BWD_MAX_BLOCKS;   // 1024: 4 workgroups / CU (512: +0.06 ms/step, 2048: +0.03)
    float* partial = static_cast<float*>(a->workspace);
    const size_t lds = (size_t)4 * 3 * a->cols * sizeof(float);
    const dim3 grid(nblk), block(256);
#define CALLF(V)                                                                                                              \
  if (xd == HERO_F32 && d == HERO_F32) hipLaunchKernelGGL((ln_bwd_fused_kernel<float, float, V>), grid, block, lds, s, *a, partial);          \
  else if (xd == HERO_F32 && d == HERO_BF16) hipLaunchKernelGGL((ln_bwd_fused_kernel<float, bf16_t, V>), grid, block, lds, s, *a, partial);   \
  else if (xd == HERO_BF16 && d == HERO_BF16) hipLaunchKernelGGL((ln_bwd_fused_kernel<bf16_t, bf16_t, V>), grid, block, lds, s, *a, partial); \
  else { set_error("hero_layernorm_bwd: unsupported dtypes x=%d dy=%d", xd, d); return HERO_ERR_UNSUPPORTED; }
    const int need = (a->cols + 255) / 256;
    if (a->cols % 256 == 0) {                          // straight-line kernel
#define CALLS(V)                                                                                                                   \
  if (xd == HERO_F32 && d == HERO_F32) hipLaunchKernelGGL((ln_bwd_fused_full_kernel<float, float, V>), grid, block, lds, s, *a, partial);          \
  else if (xd == HERO_F32 && d == HERO_BF16) hipLaunchKernelGGL((ln_bwd_fused_full_kernel<float, bf16_t, V>), grid, block, lds, s, *a, partial);   \
  else if (xd == HERO_BF16 && d == HERO_BF16) hipLaunchKernelGGL((ln_bwd_fused_full_kernel<bf16_t, bf16_t, V>), grid, block, lds, s, *a, partial); \
  else { set_error("hero_layernorm_bwd: unsupported dtypes x=%d dy=%d", xd, d); return HERO_ERR_UNSUPPORTED; }
      if (need <= 1) { CALLS(1); } else if (need <= 2) { CALLS(2); } else if (need <= 3) { CALLS(3); } else { CALLS(4); }
#undef CALLS
    } else if (need <= 1) { CALLF(1); } else if (need <= 2) { CALLF(2); } else if (need <= 3) { CALLF(3); } else { CALLF(4); }
#undef CALLF
    int rc = check_launch("hero_layernorm_bwd(fused)");
    if (rc || a->defer_fold) return rc;
    const int zs = 1;               // fixed-order fold (see run_colred)
    hipLaunchKernelGGL(colred_final3_kernel, dim3((a->cols + 63) / 64, 3, zs), dim3(256), 0, s, partial, a->dgamma, a->dbeta,
                       a->dbias_in, a->cols, nblk, a->grad_beta);
    return check_launch("hero_layernorm_bwd(final3)");
  }
  HERO_REQUIRE(!a->dbias_in, "hero_layernorm_bwd: dbias_in needs cols <= 1024");
  HERO_REQUIRE(!a->defer_fold, "hero_layernorm_bwd: defer_fold needs the fused path (cols <= 1024, dx or dbias_in wanted)");
  if (a->dx || a->dx_dropped) {
    const dim3 grid((a->rows + 3) / 4), block(256);
#define CALL(V)                                                                                                      \
  if (xd == HERO_F32 && d == HERO_F32) hipLaunchKernelGGL((ln_bwd_dx_kernel<float, float, V>), grid, block, 0, s, *a);          \
  else if (xd == HERO_F32 && d == HERO_BF16) hipLaunchKernelGGL((ln_bwd_dx_kernel<float, bf16_t, V>), grid, block, 0, s, *a);   \
  else if (xd == HERO_BF16 && d == HERO_BF16) hipLaunchKernelGGL((ln_bwd_dx_kernel<bf16_t, bf16_t, V>), grid, block, 0, s, *a); \
  else { set_error("hero_layernorm_bwd: unsupported dtypes x=%d dy=%d", xd, d); return HERO_ERR_UNSUPPORTED; }
    HERO_VPL_SWITCH(a->cols, CALL);
#undef CALL
    int rc = check_launch("hero_layernorm_bwd(dx)");
    if (rc) return rc;
  }
  if (a->dgamma || a->dbeta) {
    if (xd == HERO_F32 && d == HERO_F32)
      return run_colred<float, float>(a->x, a->dy, a->mean, a->rstd, a->dgamma, a->dbeta, a->rows, a->cols, a->cols, a->grad_beta, a->dropout_out, a->workspace, s);
    if (xd == HERO_F32 && d == HERO_BF16)
      return run_colred<float, bf16_t>(a->x, a->dy, a->mean, a->rstd, a->dgamma, a->dbeta, a->rows, a->cols, a->cols, a->grad_beta, a->dropout_out, a->workspace, s);
    if (xd == HERO_BF16 && d == HERO_BF16)
      return run_colred<bf16_t, bf16_t>(a->x, a->dy, a->mean, a->rstd, a->dgamma, a->dbeta, a->rows, a->cols, a->cols, a->grad_beta, a->dropout_out, a->workspace, s);
    set_error("hero_layernorm_bwd: unsupported dtypes x=%d dy=%d", xd, d);
    return HERO_ERR_UNSUPPORTED;
  }
  return HERO_OK;
}

extern "C" int hero_layernorm_bwd_blocks(int rows) {
  int nblk = (rows + 3) / 4;
  return nblk > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : nblk;
}

// ---- many column sums, two launches --------------------------------------------------------------------------------
// (passed by value: the whole struct must stay under the 4 KB kernel-argument limit - 64 x 48 + 772 bytes)
struct ColsumMulti {
  HeroColsum p[HERO_COLSUM_MULTI_MAX];
  int blk0[HERO_COLSUM_MULTI_MAX + 1];    // first workgroup of problem i (stage 1: col-blocks x chunks; stage 2: 16-column groups)
  int woff[HERO_COLSUM_MULTI_MAX];        // partial sums of problem i: workspace + woff[i], [nchunks][cols]
  int rpc[HERO_COLSUM_MULTI_MAX];         // rows per chunk
  int n;
};

__device__ __forceinline__ int colsum_find(const ColsumMulti& a, int b) {
  int lo = 0, hi = a.n - 1;                // uniform binary search over <= 64 prefix entries
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.blk0[mid] <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <typename T>
__device__ __forceinline__ float4 colsum_chunk(const T* src, int ld, int c, int r0, int r1, int ty) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int r = r0 + ty;
  for (; r + 12 < r1; r += 16) {           // four rows per trip, their loads issued together
    float4 d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = V4<T>::ld(src + (size_t)(r + 4 * u) * ld + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += d[u].x; s.y += d[u].y; s.z += d[u].z; s.w += d[u].w; }
  }
  for (; r < r1; r += 4) {
    const float4 d = V4<T>::ld(src + (size_t)r * ld + c);
    s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
  }
  return s;
}

__global__ __launch_bounds__(256) void colsum_multi_part_kernel(ColsumMulti a, float* ws) {
  __shared__ float4 red[4][64];
  const int pi = colsum_find(a, blockIdx.x);
  const HeroColsum P = a.p[pi];
  const int ncb = (P.cols + 255) / 256;
  const int local = blockIdx.x - a.blk0[pi], cb = local % ncb, chunk = local / ncb;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = (cb * 64 + tx) * 4;
  const int r0 = chunk * a.rpc[pi], r1 = min(P.rows, r0 + a.rpc[pi]);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < P.cols) {
    if (P.dtype == HERO_BF16) s = colsum_chunk(static_cast<const bf16_t*>(P.src), P.ld, c, r0, r1, ty);
    else s = colsum_chunk(static_cast<const float*>(P.src), P.ld, c, r0, r1, ty);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < P.cols) {
#pragma unroll
    for (int k = 1; k < 4; ++k) { const float4 v = red[k][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    *reinterpret_cast<float4*>(ws + a.woff[pi] + (size_t)chunk * P.cols + c) = s;
  }
}

// dst[c] = beta*dst[c] + sum over the chunks, fixed order: 16 columns x 16 chunk-lanes per workgroup
__global__ __launch_bounds__(256) void colsum_multi_fold_kernel(ColsumMulti a, const float* ws) {
  const int pi = colsum_find(a, blockIdx.x);
  const HeroColsum P = a.p[pi];
  const int nchunks = (P.rows + a.rpc[pi] - 1) / a.rpc[pi];
  const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
  const int c = (blockIdx.x - a.blk0[pi]) * 16 + cl;
  const float* part = ws + a.woff[pi];
  float s = 0.f;
  if (c < P.cols) {
    int k = kl;
    for (; k + 48 < nchunks; k += 64) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = part[(size_t)(k + 16 * u) * P.cols + c];
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; k < nchunks; k += 16) s += part[(size_t)k * P.cols + c];
  }
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
  __shared__ float red[4][16];
  if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][cl] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < P.cols) {
    s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (P.dst_rows) {
      // indexed destination rows (periodic position ids): distinct rows receive exactly one value per launch, so the
      // atomic is order-free; only a clamped id that repeats inside one period (positions beyond 511) shares a row
      const int j = c / P.row_cols, r = P.dst_rows[j];
      if (r >= 0) atomicAdd(P.dst + (size_t)r * P.row_cols + (c - j * P.row_cols), s);
    } else {
      P.dst[c] = (P.beta != 0.f ? P.beta * P.dst[c] : 0.f) + s;
    }
  }
}

static_assert(sizeof(ColsumMulti) <= 4000, "kernel argument size");

static int colsum_multi_plan(const HeroColsum* p, int n, ColsumMulti* a, int* fold_blocks, size_t* ws_floats) {
  size_t off = 0;
  int b1 = 0;
  for (int i = 0; i < n; ++i) {
    int rpc;
    const int nchunks = chunking(p[i].rows, &rpc);
    if (a) { a->p[i] = p[i]; a->blk0[i] = b1; a->woff[i] = (int)off; a->rpc[i] = rpc; }
    b1 += ((p[i].cols + 255) / 256) * nchunks;
    off += (size_t)nchunks * p[i].cols;
  }
  if (a) { a->blk0[n] = b1; a->n = n; }
  if (fold_blocks) {
    int b2 = 0;
    for (int i = 0; i < n; ++i) b2 += (p[i].cols + 15) / 16;
    *fold_blocks = b2;
  }
  *ws_floats = off;
  return b1;
}

extern "C" size_t hero_colsum_multi_workspace_bytes(const HeroColsum* p, int n) {
  if (!p || n < 1 || n > HERO_COLSUM_MULTI_MAX) return 0;
  size_t fl = 0;
  colsum_multi_plan(p, n, nullptr, nullptr, &fl);
  return fl * sizeof(float);
}

extern "C" int hero_colsum_multi(const HeroColsum* p, int n, void* workspace, hero_stream_t stream) {
  HERO_REQUIRE(p && workspace && n >= 1 && n <= HERO_COLSUM_MULTI_MAX, "hero_colsum_multi: 1..%d problems and a workspace", HERO_COLSUM_MULTI_MAX);
  for (int i = 0; i < n; ++i) {
    HERO_REQUIRE(p[i].src && p[i].dst && p[i].rows > 0 && p[i].cols > 0 && p[i].cols % 4 == 0 && p[i].ld % 4 == 0 &&
                     (p[i].dtype == HERO_BF16 || p[i].dtype == HERO_F32) && ((uintptr_t)p[i].src & 7) == 0,
                 "hero_colsum_multi: bad problem %d", i);
    HERO_REQUIRE(!p[i].dst_rows || (p[i].beta == 1.f && p[i].row_cols > 0 && p[i].cols % p[i].row_cols == 0),
                 "hero_colsum_multi: problem %d: dst_rows needs beta = 1 and cols a multiple of row_cols", i);
  }
  ColsumMulti a, f;
  size_t fl = 0;
  int fold_blocks = 0;
  const int part_blocks = colsum_multi_plan(p, n, &a, &fold_blocks, &fl);
  HERO_REQUIRE(fl < 0x7fffffffull, "hero_colsum_multi: workspace too large");
  f = a;                                    // stage 2 walks 16-column groups instead
  int b2 = 0;
  for (int i = 0; i < n; ++i) { f.blk0[i] = b2; b2 += (p[i].cols + 15) / 16; }
  f.blk0[n] = b2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(colsum_multi_part_kernel, dim3(part_blocks), dim3(256), 0, s, a, static_cast<float*>(workspace));
  int rc = check_launch("hero_colsum_multi(partials)");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_multi_fold_kernel, dim3(fold_blocks), dim3(256), 0, s, f, static_cast<const float*>(workspace));
  return check_launch("hero_colsum_multi(fold)");
}

extern "C" int hero_colsum(const void* x, float* out, int rows, int cols, int ld, int dtype, float beta, void* workspace,
                           hero_stream_t stream) {
  HERO_REQUIRE(x && out && workspace, "hero_colsum: null pointer");
  HERO_REQUIRE(cols > 0 && cols % 4 == 0 && ld % 4 == 0, "hero_colsum: cols/ld (%d, %d) must be multiples of 4", cols, ld);
  hipStream_t s = static_cast<hipStream_t>(stream);
  HeroDropout none = {nullptr, 0, 0, 1.f};
  if (rows <= 0) {
    if (beta == 0.f) hipLaunchKernelGGL(colred_final_kernel, dim3((cols + 15) / 16), dim3(256), 0, s, nullptr, nullptr, nullptr, out, cols, 0, 0.f);   // zeros (a kernel node, not a memset)
    return HERO_OK;
  }
  if (dtype == HERO_F32) return run_colred<float, float>(nullptr, x, nullptr, nullptr, nullptr, out, rows, cols, ld, beta, none, workspace, s);
  if (dtype == HERO_BF16) return run_colred<float, bf16_t>(nullptr, x, nullptr, nullptr, nullptr, out, rows, cols, ld, beta, none, workspace, s);
  set_error("hero_colsum: bad dtype %d", dtype);
  return HERO_ERR_ARG;
}
