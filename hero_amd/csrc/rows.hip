// Row gathers / scatters, elementwise helpers and the fused optimiser (all HBM-bound, gfx950).
#include "common.h"

namespace hero {

// ---- out[r] = a[idx] | 0 | b[-idx-2] -----------------------------------------------------------
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ a, const T* __restrict__ b, const int32_t* __restrict__ idx,
                                   T* __restrict__ out, int rows, int cols) {
  const int c4n = cols >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)rows * c4n; q += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(q / c4n), c = (int)(q - (size_t)r * c4n) * 4;
    const int i = idx[r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 0) v = V4<T>::ld(a + (size_t)i * cols + c);
    else if (i <= -2) v = V4<T>::ld(b + (size_t)(-i - 2) * cols + c);
    V4<T>::st(out + (size_t)r * cols + c, v);
  }
}

// ---- first occurrence of every source row in a gather index ----------------------------------------
// inv[r] (a rows) / inv[na + r] (b rows) = the smallest output row j whose idx[j] names that source row, -1 if none.  One
// workgroup, the whole table in LDS (atomicMin is order-independent: deterministic).
__global__ __launch_bounds__(1024) void inverse_first_kernel(const int32_t* __restrict__ idx, int n, int32_t* __restrict__ inv, int na, int nb) {
  extern __shared__ int lds_inv[];
  const int tot = na + nb;
  for (int i = threadIdx.x; i < tot; i += 1024) lds_inv[i] = 0x7fffffff;
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 1024) {
    const int v = idx[j];
    if (v >= 0 && v < na) atomicMin(&lds_inv[v], j);
    else if (v <= -2 && -v - 2 < nb) atomicMin(&lds_inv[na - v - 2], j);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tot; i += 1024) { const int v = lds_inv[i]; inv[i] = v == 0x7fffffff ? -1 : v; }
}

// ---- out[r] = sum over CSR entries ---------------------------------------------------------------
template <typename T>
__global__ void csr_gather_sum_kernel(const T* __restrict__ src, const int32_t* __restrict__ offs,
                                      const int32_t* __restrict__ ent, T* __restrict__ out, int rows, int cols) {
  const int c4n = cols >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)rows * c4n; q += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(q / c4n), c = (int)(q - (size_t)r * c4n) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = offs[r]; e < offs[r + 1]; ++e) {
      const float4 w = V4<T>::ld(src + (size_t)ent[e] * cols + c);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    V4<T>::st(out + (size_t)r * cols + c, v);
  }
}

// ---- dst[idx[r]] += src[r] -------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add2(float* p, float x, float y) {
  atomicAdd(p, x);
  atomicAdd(p + 1, y);
}
__device__ __forceinline__ void atomic_add2(bf16_t* p, float x, float y) {  // p is 4-byte aligned (even column)
  uint32_t* w = reinterpret_cast<uint32_t*>(p);
  uint32_t old = *w, assumed;
  do {
    assumed = old;
    const float lo = __uint_as_float(assumed << 16) + x;
    const float hi = __uint_as_float(assumed & 0xffff0000u) + y;
    const uint32_t nw = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    old = atomicCAS(w, assumed, nw);
  } while (old != assumed);
}

template <typename TS, typename TD>
__global__ void scatter_add_rows_kernel(const TS* __restrict__ src, const int32_t* __restrict__ idx, TD* dst_a, TD* dst_b,
                                        int rows, int cols, int skip) {
  const int c2n = cols >> 1;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < (size_t)rows * c2n; q += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(q / c2n), c = (int)(q - (size_t)r * c2n) * 2;
    const int i = idx[r];
    if (i == -1 || i == skip) continue;
    const float x = ld1<TS>(src + (size_t)r * cols + c), y = ld1<TS>(src + (size_t)r * cols + c + 1);
    if (x == 0.f && y == 0.f) continue;
    TD* d = i >= 0 ? dst_a + (size_t)i * cols + c : (dst_b ? dst_b + (size_t)(-i - 2) * cols + c : nullptr);
    if (d) atomic_add2(d, x, y);
  }
}

// ---- deterministic embedding-table gradients: sort the rows by destination, one owner per destination ----------
// dst[idx[r]] += src[r] with fp32 atomics depends on the order the atomics land in as soon as a destination receives
// three or more rows (frequent tokens: every real batch; ~100 ids of the synthetic one) - the last kernel of the 1-GPU
// step whose result was not reproducible run to run.  Instead:
//   segment_sort_kernel   ONE workgroup, stable LSD radix sort (4-bit digits) of (destination, row) pairs: 1024 threads own
//                         contiguous chunks of the list, count their 16 digits, an exclusive scan over [digit][thread]
//                         gives every thread its write positions, elements move in chunk order (stable).  The list is
//                         ~10^4 ids per micro-step (10^6 at config 5: ~1 ms of a 600 ms step); it is sorted once per
//                         batch object (functional.memo) and refreshed with the batch by the feeder's commit graph.
//   scatter_sorted_kernel one wave per sorted position; the head of a run of equal destinations sums the run's source
//                         rows in row order (fp32) and adds the sum to the table row it alone owns - no atomics.
constexpr int SORT_NT = 1024;
__global__ __launch_bounds__(SORT_NT) void segment_sort_kernel(const int32_t* __restrict__ idx, int rows, int skip, int nbits,
                                                               uint2* __restrict__ buf0, uint2* __restrict__ buf1,
                                                               int32_t* __restrict__ order) {
  extern __shared__ uint32_t hist[];                 // [16][SORT_NT] counters / offsets, then [32] wave totals
  uint32_t* wsum = hist + 16 * SORT_NT;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int cpt = (rows + SORT_NT - 1) / SORT_NT;
  const int beg = min(t * cpt, rows), end = min(beg + cpt, rows);
  const uint32_t last_key = (1u << nbits) - 1u;      // dropped rows (idx < 0 or == skip) sort behind every real destination
  for (int i = beg; i < end; ++i) {
    const int v = idx[i];
    buf0[i] = make_uint2((v < 0 || v == skip) ? last_key : (uint32_t)v, (uint32_t)i);
  }
  __syncthreads();
  uint2* src = buf0;
  uint2* dst = buf1;
  for (int shift = 0; shift < nbits; shift += 4) {
    uint32_t cnt[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) cnt[b] = 0;
    for (int i = beg; i < end; ++i) {
      const uint32_t d = (src[i].x >> shift) & 15u;
#pragma unroll
      for (int b = 0; b < 16; ++b) cnt[b] += (d == (uint32_t)b);
    }
#pragma unroll
    for (int b = 0; b < 16; ++b) hist[b * SORT_NT + t] = cnt[b];
    __syncthreads();
    // exclusive scan of the flattened [digit][thread] array: thread t owns entries 16 t .. 16 t + 15
    uint32_t loc[16], run = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { loc[k] = run; run += hist[16 * t + k]; }
    uint32_t inc = run;                               // inclusive scan of `run` over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t base = inc - run;
    for (int w = 0; w < wv; ++w) base += wsum[w];
#pragma unroll
    for (int k = 0; k < 16; ++k) hist[16 * t + k] = base + loc[k];
    __syncthreads();
    uint32_t pos[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) pos[b] = hist[b * SORT_NT + t];
    for (int i = beg; i < end; ++i) {
      const uint2 e = src[i];
      const uint32_t d = (e.x >> shift) & 15u;
      uint32_t at = 0;
#pragma unroll
      for (int b = 0; b < 16; ++b)
        if (d == (uint32_t)b) { at = pos[b]; pos[b] += 1; }
      dst[at] = e;
    }
    __syncthreads();                                   // (global writes of this workgroup are visible to it behind the barrier)
    uint2* tmp = src; src = dst; dst = tmp;
  }
  for (int i = t; i < rows; i += SORT_NT) order[i] = (int32_t)src[i].y;
}

// Blocks of SEG_B sorted positions, one wave each.  A run of equal destinations that lies inside one block is summed and
// added to its table row by that wave (the row has no other writer).  A run that crosses block boundaries - the SEP
// token heads every subtitle: 480 rows of the TVR batch go to ONE table row - leaves one partial sum per block in the
// workspace (slot 0: the block's first run continues from the previous block, slot 1: its last run continues into the
// next one), and the second kernel lets the wave of the run's FIRST block add the partials up in block order.
#ifndef HERO_SEG_B
#define HERO_SEG_B 16            // round 6 (tools/lab/scatter_time.py, us per call: 9600 sub-tokens / the MLM batch / 480 query tokens):
                                 // 32: 28.7 / 32.2 / 20.2   16: 24.1 / 31.0 / 13.5   8: 26.1 / 39.6 / 9.5 - more, shorter blocks put more waves in
                                 // flight; at 8 the fold of the <mask> run (1400 rows of one id) spans too many blocks
#endif
constexpr int SEG_B = HERO_SEG_B;
constexpr int UNR = 8;        // source rows (and their table rows) in flight per wave (4: 32 us per call on the bench batch, eight round trips per wave)
__device__ __forceinline__ int seg_key(const int32_t* idx, const int32_t* order, int i, int rows, int skip) {
  if (i < 0 || i >= rows) return -2;
  const int v = idx[order[i]];
  return (v < 0 || v == skip) ? -1 : v;
}

template <typename TS, int NCH>    // NCH chunks of 256 columns per lane pass (cols <= NCH * 256 per outer iteration)
__global__ __launch_bounds__(256) void scatter_sorted_kernel(const TS* __restrict__ src, const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ order, float* __restrict__ dst,
                                                             float* __restrict__ partial, int rows, int cols, int skip) {
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int base = blk * SEG_B;
  if (base >= rows) return;
  const int n = min(SEG_B, rows - base);
  // lane p < n: row number and key of sorted position base + p
  const int my_row = lane < n ? order[base + lane] : 0;
  int my_key = -2;
  if (lane < n) { const int v = idx[my_row]; my_key = (v < 0 || v == skip) ? -1 : v; }
  const int prev_key = seg_key(idx, order, base - 1, rows, skip), next_key = seg_key(idx, order, base + n, rows, skip);
  for (int c0 = 0; c0 < cols; c0 += NCH * 256) {
    float4 acc[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    int seg_start = 0;
    for (int p0 = 0; p0 < n; p0 += UNR) {
      // UNR rows in flight, and with them the table rows they may be added to: the loads only depend on the shuffled row
      // numbers / keys, so a group costs ONE memory round trip (read-add-write per run, one after the other, was a round
      // trip per run: 64 us for the 9600 sub-tokens of the bench batch, most of them runs of one)
      float4 v[UNR][NCH], old[UNR][NCH];
      int keys[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int p = min(p0 + u, n - 1);
        keys[u] = __shfl(my_key, p, 64);
        const int r = __shfl(my_row, p, 64);
        const int kd = max(keys[u], 0);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const int c = min(c0 + k * 256 + lane * 4, cols - 4);
          v[u][k] = V4<TS>::ld(src + (size_t)r * cols + c);
          old[u][k] = *reinterpret_cast<const float4*>(dst + (size_t)kd * cols + c);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int p = p0 + u;
        if (p < n) {
          const int key = keys[u];
          const int key_next = p + 1 < n ? __shfl(my_key, p + 1, 64) : -3;
          if (key >= 0) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) { acc[k].x += v[u][k].x; acc[k].y += v[u][k].y; acc[k].z += v[u][k].z; acc[k].w += v[u][k].w; }
          }
          if (key_next != key) {                         // the run [seg_start, p] ends here (inside the block or at its end)
            if (key >= 0) {
              const bool left_open = seg_start == 0 && prev_key == key;
              const bool right_open = p == n - 1 && next_key == key;
              const bool closed = !left_open && !right_open;
              float* out = closed ? dst + (size_t)key * cols : partial + ((size_t)blk * 2 + (left_open ? 0 : 1)) * cols;
#pragma unroll
              for (int k = 0; k < NCH; ++k) {
                const int c = c0 + k * 256 + lane * 4;
                if (c < cols) {
                  float4 o = acc[k];
                  if (closed) { o.x += old[u][k].x; o.y += old[u][k].y; o.z += old[u][k].z; o.w += old[u][k].w; }   // old[u]: this run's table row
                  *reinterpret_cast<float4*>(out + c) = o;
                }
              }
            }
#pragma unroll
            for (int k = 0; k < NCH; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            seg_start = p + 1;
          }
        }
      }
    }
  }
}

// the wave of the block in which a multi-block run STARTS folds the run's partial sums in block order
__global__ __launch_bounds__(256) void scatter_sorted_fold_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ order,
                                                                  float* __restrict__ dst, const float* __restrict__ partial,
                                                                  int rows, int cols, int skip) {
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int base = blk * SEG_B;
  if (base >= rows) return;
  const int n = min(SEG_B, rows - base);
  const int last = seg_key(idx, order, base + n - 1, rows, skip);
  if (last < 0 || seg_key(idx, order, base + n, rows, skip) != last) return;        // the last run does not continue
  // it continues; is this block the run's first?  (not if the whole block is the run and it came in from the left)
  const bool whole = seg_key(idx, order, base, rows, skip) == last;
  if (whole && seg_key(idx, order, base - 1, rows, skip) == last) return;
  const int nblk = (rows + SEG_B - 1) / SEG_B;
  // How far the run reaches: blocks blk + 1 ... blk + nb start with the run's key (each left its share in slot 0).  64 blocks
  // are probed per step, one per lane (the serial walk - two dependent index loads and a partial-sum load per block, once
  // per 256-column chunk - was 18 us for the SEP token's 15 blocks and 50 us for the <mask> token's 45 of the MLM batch).
  int nb = 0;
  for (int b0 = blk + 1; b0 < nblk; b0 += 64) {
    const int b = b0 + lane;
    const bool cont = b < nblk && seg_key(idx, order, b * SEG_B, rows, skip) == last;
    const unsigned long long m = __ballot(cont);
    const int run = m == ~0ull ? 64 : __builtin_ctzll(~m);
    nb += run;
    if (run < 64) break;
  }
  constexpr int NCH = 3, FB = 8;                       // 3 x 256 columns per lane pass, 8 partial rows in flight
  for (int c0 = 0; c0 < cols; c0 += NCH * 256) {
    float4 acc[NCH];
    int cc[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      cc[k] = min(c0 + k * 256 + lane * 4, cols - 4);
      acc[k] = *reinterpret_cast<const float4*>(partial + ((size_t)blk * 2 + 1) * cols + cc[k]);
    }
    for (int b0 = 1; b0 <= nb; b0 += FB) {
      float4 t[FB][NCH];
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int b = blk + min(b0 + u, nb);
#pragma unroll
        for (int k = 0; k < NCH; ++k) t[u][k] = *reinterpret_cast<const float4*>(partial + ((size_t)b * 2 + 0) * cols + cc[k]);
      }
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        if (b0 + u <= nb) {                            // uniform; block order = the order of the serial walk
#pragma unroll
          for (int k = 0; k < NCH; ++k) { acc[k].x += t[u][k].x; acc[k].y += t[u][k].y; acc[k].z += t[u][k].z; acc[k].w += t[u][k].w; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = c0 + k * 256 + lane * 4;
      if (c < cols) {
        float4* d = reinterpret_cast<float4*>(dst + (size_t)last * cols + c);
        float4 o = *d;
        o.x += acc[k].x; o.y += acc[k].y; o.z += acc[k].z; o.w += acc[k].w;
        *d = o;
      }
    }
  }
}

// ---- elementwise -------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, size_t n) {
  const size_t n4 = n >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x)
    V4<TD>::st(d + q * 4, V4<TS>::ld(s + q * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) st1<TD>(d + n4 * 4 + threadIdx.x, ld1<TS>(s + n4 * 4 + threadIdx.x));
}
// dst[c * ldd + r] = (TD) src[r * C + c]   (32x32 tiles through LDS, both sides coalesced)
template <typename TD>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, TD* __restrict__ dst, int R, int C, int ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < R) st1<TD>(dst + (size_t)c * ldd + r, tile[tx][ty + 8 * k]);
  }
}

// one 64 x 64 tile of one tensor per workgroup (see hero_copy_multi)
__global__ __launch_bounds__(256) void copy_multi_kernel(const HeroCopyDesc* __restrict__ descs, const int32_t* __restrict__ tile_desc,
                                                         const int32_t* __restrict__ tile_index) {
  __shared__ float tile[64][65];
  const HeroCopyDesc d = descs[tile_desc[blockIdx.x]];
  const int tcols = (d.cols + 63) >> 6;
  const int ti = tile_index[blockIdx.x];
  const int r0 = (ti / tcols) * 64, c0 = (ti % tcols) * 64;
  // 16 float4 per tile row: thread (q = tid & 15, ty = tid >> 4) moves 4 consecutive columns of rows ty, ty + 16, ...
  // (16-byte loads, 8-byte bf16 stores; the scalar 4-byte / 2-byte version ran at 2.4 TB/s)
  const int q = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16
  const bool vec = (d.cols & 3) == 0 && (d.ldd & 3) == 0 && ((uintptr_t)d.src & 15) == 0 && ((uintptr_t)d.dst & 15) == 0;
  if (!d.transpose) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + ty + 16 * k, c = c0 + 4 * q;
      if (r >= d.rows || c >= d.cols) continue;
      if (vec) {                                               // cols % 4 == 0: c + 3 < cols
        const float4 v = *reinterpret_cast<const float4*>(d.src + (size_t)r * d.cols + c);
        if (d.dst_dtype == HERO_BF16) V4<bf16_t>::st(static_cast<bf16_t*>(d.dst) + (size_t)r * d.ldd + c, v);
        else *reinterpret_cast<float4*>(static_cast<float*>(d.dst) + (size_t)r * d.ldd + c) = v;
      } else {
        for (int e = 0; e < 4 && c + e < d.cols; ++e) {
          const float v = d.src[(size_t)r * d.cols + c + e];
          if (d.dst_dtype == HERO_BF16) static_cast<bf16_t*>(d.dst)[(size_t)r * d.ldd + c + e] = f2bf(v);
          else static_cast<float*>(d.dst)[(size_t)r * d.ldd + c + e] = v;
        }
      }
    }
    return;
  }
  const bool vin = (d.cols & 3) == 0 && ((uintptr_t)d.src & 15) == 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rl = ty + 16 * k, r = r0 + rl, c = c0 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < d.rows && c < d.cols) {
      if (vin) v = *reinterpret_cast<const float4*>(d.src + (size_t)r * d.cols + c);
      else {
        v.x = d.src[(size_t)r * d.cols + c];
        if (c + 1 < d.cols) v.y = d.src[(size_t)r * d.cols + c + 1];
        if (c + 2 < d.cols) v.z = d.src[(size_t)r * d.cols + c + 2];
        if (c + 3 < d.cols) v.w = d.src[(size_t)r * d.cols + c + 3];
      }
    }
    tile[rl][4 * q] = v.x; tile[rl][4 * q + 1] = v.y; tile[rl][4 * q + 2] = v.z; tile[rl][4 * q + 3] = v.w;
  }
  __syncthreads();
  // output row = source column c, 4 consecutive source rows per thread
  const bool vout = (d.ldd & 3) == 0 && (((uintptr_t)d.dst) & 15) == 0 && (d.rows & 3) == 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cl = ty + 16 * k, c = c0 + cl, r = r0 + 4 * q;
    if (c >= d.cols || r >= d.rows) continue;
    const float4 v = make_float4(tile[4 * q][cl], tile[4 * q + 1][cl], tile[4 * q + 2][cl], tile[4 * q + 3][cl]);
    if (vout) {                                                // rows % 4 == 0: r + 3 < rows
      if (d.dst_dtype == HERO_BF16) V4<bf16_t>::st(static_cast<bf16_t*>(d.dst) + (size_t)c * d.ldd + r, v);
      else *reinterpret_cast<float4*>(static_cast<float*>(d.dst) + (size_t)c * d.ldd + r) = v;
    } else {
      const float e4[4] = {v.x, v.y, v.z, v.w};
      for (int e = 0; e < 4 && r + e < d.rows; ++e) {
        if (d.dst_dtype == HERO_BF16) static_cast<bf16_t*>(d.dst)[(size_t)c * d.ldd + r + e] = f2bf(e4[e]);
        else static_cast<float*>(d.dst)[(size_t)c * d.ldd + r + e] = e4[e];
      }
    }
  }
}

template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, size_t n) {
  const size_t n4 = n >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    float4 g = V4<T>::ld(dy + q * 4);
    const float4 v = V4<T>::ld(y + q * 4);
    g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
    V4<T>::st(dx + q * 4, g);
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u, T* __restrict__ dx, size_t n) {
  const size_t n4 = n >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    float4 g = V4<T>::ld(dy + q * 4);
    const float4 v = V4<T>::ld(u + q * 4);
    g.x *= gelu_erf_grad(v.x); g.y *= gelu_erf_grad(v.y); g.z *= gelu_erf_grad(v.z); g.w *= gelu_erf_grad(v.w);
    V4<T>::st(dx + q * 4, g);
  }
}
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t n) {
  const size_t n4 = n >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    float4 u = V4<T>::ld(a + q * 4);
    const float4 v = V4<T>::ld(b + q * 4);
    u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
    V4<T>::st(y + q * 4, u);
  }
}

// ---- optimiser -----------------------------------------------------------------------------------------
__global__ void sumsq_kernel(const float* __restrict__ g, size_t n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  const size_t n4 = n >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(g + q * 4);
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);   // per-block partial
}
// fixed-order fold of the partials: the gradient norm is bit-identical on every data-parallel rank
// (an atomicAdd fold made replicas drift apart by an ulp per step through the clipping factor)
__global__ void sumsq_final_kernel(const float* __restrict__ partial, int nblocks, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void adamw_kernel(HeroAdamW a, float bc1, float bc2) {
  float gs = a.grad_scale;
  if (a.grad_sumsq) {
    const float norm = sqrtf(*a.grad_sumsq) * fabsf(a.grad_scale);
    const float clip = a.max_grad_norm / (norm + 1e-6f);
    if (clip < 1.f) gs *= clip;
  }
  const float step_size = a.lr * sqrtf(bc2) / bc1;
  const float decay = a.lr * a.weight_decay;
  bf16_t* sh = static_cast<bf16_t*>(a.shadow);
  const size_t n4 = a.n >> 2;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    float4 p = *reinterpret_cast<float4*>(a.p + q * 4);
    const float4 g4 = *reinterpret_cast<const float4*>(a.g + q * 4);
    float4 m = *reinterpret_cast<float4*>(a.m + q * 4);
    float4 v = *reinterpret_cast<float4*>(a.v + q * 4);
    float* pp = &p.x; const float* gp = &g4.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = gp[k] * gs;
      mp[k] = a.beta1 * mp[k] + (1.f - a.beta1) * g;
      vp[k] = a.beta2 * vp[k] + (1.f - a.beta2) * g * g;
      pp[k] -= step_size * mp[k] / (sqrtf(vp[k]) + a.eps);
      pp[k] -= decay * pp[k];
    }
    *reinterpret_cast<float4*>(a.p + q * 4) = p;
    *reinterpret_cast<float4*>(a.m + q * 4) = m;
    *reinterpret_cast<float4*>(a.v + q * 4) = v;
    if (sh) V4<bf16_t>::st(sh + q * 4, p);
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float g = a.g[i] * gs;
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    float p = a.p[i] - step_size * m / (sqrtf(v) + a.eps);
    p -= decay * p;
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
    if (sh) sh[i] = f2bf(p);
  }
}

// One launch over many tensors (descriptor tables live in device memory, one block per 16K-element
// chunk).  Same arithmetic as adamw_kernel.
constexpr int MT_CHUNK = 16384;
__global__ void adamw_steps_inc_kernel(const HeroTensorDesc* descs, int n, int32_t* steps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) steps[descs[i].step_lag] += 1;            // slots are distinct per tensor
}
__global__ __launch_bounds__(256) void adamw_multi_kernel(HeroAdamWMulti a) {
  const int ti = a.chunk_tensor[blockIdx.x];
  const HeroTensorDesc d = a.descs[ti];
  const HeroAdamWGroup gr = a.groups[d.group];
  const size_t beg = (size_t)a.chunk_index[blockIdx.x] * MT_CHUNK;
  const size_t end = beg + MT_CHUNK < d.n ? beg + MT_CHUNK : d.n;
  float gs = a.grad_scale;
  if (a.grad_sumsq) {
    const float norm = sqrtf(*a.grad_sumsq) * fabsf(a.grad_scale);
    const float clip = a.max_grad_norm / (norm + 1e-6f);
    if (clip < 1.f) gs *= clip;
  }
  const float st = a.tensor_steps ? (float)a.tensor_steps[d.step_lag] : (float)((a.step_ptr ? *a.step_ptr : a.step) - d.step_lag);
  const float lr = a.lr_ptr ? a.lr_ptr[d.group] : gr.lr;
  const float bc1 = 1.f - powf(gr.beta1, st), bc2 = 1.f - powf(gr.beta2, st);
  const float step_size = lr * sqrtf(bc2) / bc1;
  const float decay = lr * gr.weight_decay;
  const bool vec = (((uintptr_t)d.p | (uintptr_t)d.g | (uintptr_t)d.m | (uintptr_t)d.v) & 15) == 0;
  if (vec) {
    const size_t e4 = beg + ((end - beg) & ~(size_t)3);
    for (size_t i = beg + threadIdx.x * 4; i < e4; i += 256 * 4) {
      float4 p = *reinterpret_cast<float4*>(d.p + i);
      const float4 g4 = *reinterpret_cast<const float4*>(d.g + i);
      float4 m = *reinterpret_cast<float4*>(d.m + i);
      float4 v = *reinterpret_cast<float4*>(d.v + i);
      float* pp = &p.x; const float* gp = &g4.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = gp[k] * gs;
        mp[k] = gr.beta1 * mp[k] + (1.f - gr.beta1) * g;
        vp[k] = gr.beta2 * vp[k] + (1.f - gr.beta2) * g * g;
        pp[k] -= step_size * mp[k] / (sqrtf(vp[k]) + gr.eps);
        pp[k] -= decay * pp[k];
      }
      *reinterpret_cast<float4*>(d.p + i) = p;
      *reinterpret_cast<float4*>(d.m + i) = m;
      *reinterpret_cast<float4*>(d.v + i) = v;
    }
    for (size_t i = e4 + threadIdx.x; i < end; i += 256) {
      const float g = d.g[i] * gs;
      const float m = gr.beta1 * d.m[i] + (1.f - gr.beta1) * g;
      const float v = gr.beta2 * d.v[i] + (1.f - gr.beta2) * g * g;
      float p = d.p[i] - step_size * m / (sqrtf(v) + gr.eps);
      p -= decay * p;
      d.p[i] = p; d.m[i] = m; d.v[i] = v;
    }
  } else {
    for (size_t i = beg + threadIdx.x; i < end; i += 256) {
      const float g = d.g[i] * gs;
      const float m = gr.beta1 * d.m[i] + (1.f - gr.beta1) * g;
      const float v = gr.beta2 * d.v[i] + (1.f - gr.beta2) * g * g;
      float p = d.p[i] - step_size * m / (sqrtf(v) + gr.eps);
      p -= decay * p;
      d.p[i] = p; d.m[i] = m; d.v[i] = v;
    }
  }
}

static inline int grid_for(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

}  // namespace hero

using namespace hero;

extern "C" int hero_gather_rows(const void* a, const void* b, const int32_t* idx, void* out, int rows, int cols, int dtype,
                                hero_stream_t stream) {
  HERO_REQUIRE(idx && out, "hero_gather_rows: null pointer");
  HERO_REQUIRE(cols > 0 && cols % 4 == 0, "hero_gather_rows: cols (%d) must be a multiple of 4", cols);
  if (rows <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for((size_t)rows * (cols >> 2));
  if (dtype == HERO_F32)
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, idx, (float*)out, rows, cols);
  else if (dtype == HERO_BF16)
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, idx, (bf16_t*)out, rows, cols);
  else { set_error("hero_gather_rows: bad dtype %d", dtype); return HERO_ERR_ARG; }
  return check_launch("hero_gather_rows");
}

extern "C" int hero_inverse_first(const int32_t* idx, int n, int32_t* inv, int na, int nb, hero_stream_t stream) {
  HERO_REQUIRE(idx && inv && n >= 0 && na >= 0 && nb >= 0, "hero_inverse_first: bad arguments");
  if (na + nb == 0) return HERO_OK;
  const size_t lds = (size_t)(na + nb) * sizeof(int);
  if (lds > 150 * 1024) { set_error("hero_inverse_first: %d source rows exceed the LDS-resident table (38400)", na + nb); return HERO_ERR_UNSUPPORTED; }
  if (lds > 65536) HERO_ENSURE_LDS(&inverse_first_kernel, 150 * 1024, "inverse_first_kernel");
  hipLaunchKernelGGL(inverse_first_kernel, dim3(1), dim3(1024), lds, static_cast<hipStream_t>(stream), idx, n, inv, na, nb);
  return check_launch("hero_inverse_first");
}

extern "C" int hero_csr_gather_sum(const void* src, const int32_t* offsets, const int32_t* entries, void* out, int rows,
                                   int cols, int dtype, hero_stream_t stream) {
  HERO_REQUIRE(src && offsets && out, "hero_csr_gather_sum: null pointer");
  HERO_REQUIRE(cols > 0 && cols % 4 == 0, "hero_csr_gather_sum: cols (%d) must be a multiple of 4", cols);
  if (rows <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for((size_t)rows * (cols >> 2));
  if (dtype == HERO_F32)
    hipLaunchKernelGGL(csr_gather_sum_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, offsets, entries, (float*)out, rows, cols);
  else if (dtype == HERO_BF16)
    hipLaunchKernelGGL(csr_gather_sum_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)src, offsets, entries, (bf16_t*)out, rows, cols);
  else { set_error("hero_csr_gather_sum: bad dtype %d", dtype); return HERO_ERR_ARG; }
  return check_launch("hero_csr_gather_sum");
}

extern "C" int hero_scatter_add_rows(const void* src, const int32_t* idx, void* dst_a, void* dst_b, int rows, int cols,
                                     int src_dtype, int dst_dtype, int skip_idx, hero_stream_t stream) {
  HERO_REQUIRE(src && idx && dst_a, "hero_scatter_add_rows: null pointer");
  HERO_REQUIRE(cols > 0 && cols % 2 == 0, "hero_scatter_add_rows: cols (%d) must be even", cols);
  if (rows <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for((size_t)rows * (cols >> 1));
  const int skip = skip_idx >= 0 ? skip_idx : -1;
  if (src_dtype == HERO_F32 && dst_dtype == HERO_F32)
    hipLaunchKernelGGL((scatter_add_rows_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, idx, (float*)dst_a, (float*)dst_b, rows, cols, skip);
  else if (src_dtype == HERO_BF16 && dst_dtype == HERO_F32)
    hipLaunchKernelGGL((scatter_add_rows_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, idx, (float*)dst_a, (float*)dst_b, rows, cols, skip);
  else if (src_dtype == HERO_BF16 && dst_dtype == HERO_BF16)
    hipLaunchKernelGGL((scatter_add_rows_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, idx, (bf16_t*)dst_a, (bf16_t*)dst_b, rows, cols, skip);
  else { set_error("hero_scatter_add_rows: unsupported dtypes %d -> %d", src_dtype, dst_dtype); return HERO_ERR_UNSUPPORTED; }
  return check_launch("hero_scatter_add_rows");
}

extern "C" size_t hero_segment_sort_workspace_bytes(int rows) { return (size_t)(rows > 0 ? rows : 0) * 2 * sizeof(uint2); }

extern "C" int hero_segment_sort(const int32_t* idx, int rows, int n_dst, int skip_idx, int32_t* order, void* workspace,
                                 hero_stream_t stream) {
  HERO_REQUIRE(idx && order && workspace, "hero_segment_sort: null pointer");
  HERO_REQUIRE(n_dst > 0 && n_dst < (1 << 30), "hero_segment_sort: n_dst = %d", n_dst);
  if (rows <= 0) return HERO_OK;
  int nbits = 1;
  while ((1 << nbits) <= n_dst) ++nbits;              // keys 0 .. n_dst - 1 and the all-ones key of the dropped rows
  nbits = (nbits + 3) & ~3;
  const int lds = (16 * SORT_NT + 32) * (int)sizeof(uint32_t);
  HERO_ENSURE_LDS(&segment_sort_kernel, lds, "segment_sort_kernel");
  uint2* b0 = static_cast<uint2*>(workspace);
  hipLaunchKernelGGL(segment_sort_kernel, dim3(1), dim3(SORT_NT), lds, static_cast<hipStream_t>(stream), idx, rows,
                     skip_idx >= 0 ? skip_idx : -1, nbits, b0, b0 + rows, order);
  return check_launch("hero_segment_sort");
}

extern "C" size_t hero_scatter_add_sorted_workspace_bytes(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (size_t)((rows + SEG_B - 1) / SEG_B) * 2 * (size_t)cols * sizeof(float);
}

extern "C" int hero_scatter_add_sorted(const void* src, const int32_t* idx, const int32_t* order, float* dst, int rows, int cols,
                                       int src_dtype, int skip_idx, void* workspace, hero_stream_t stream) {
  HERO_REQUIRE(src && idx && order && dst && workspace, "hero_scatter_add_sorted: null pointer");
  HERO_REQUIRE(cols > 0 && cols % 4 == 0, "hero_scatter_add_sorted: cols (%d) must be a multiple of 4", cols);
  if (rows <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nblk = (rows + SEG_B - 1) / SEG_B, grid = (nblk + 3) / 4, skip = skip_idx >= 0 ? skip_idx : -1;
  float* part = static_cast<float*>(workspace);
  if (src_dtype == HERO_BF16) hipLaunchKernelGGL((scatter_sorted_kernel<bf16_t, 3>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, idx, order, dst, part, rows, cols, skip);
  else if (src_dtype == HERO_F32) hipLaunchKernelGGL((scatter_sorted_kernel<float, 3>), dim3(grid), dim3(256), 0, s, (const float*)src, idx, order, dst, part, rows, cols, skip);
  else { set_error("hero_scatter_add_sorted: bad dtype %d", src_dtype); return HERO_ERR_ARG; }
  int rc = check_launch("hero_scatter_add_sorted");
  if (rc) return rc;
  hipLaunchKernelGGL(scatter_sorted_fold_kernel, dim3(grid), dim3(256), 0, s, idx, order, dst, part, rows, cols, skip);
  return check_launch("hero_scatter_add_sorted(fold)");
}

extern "C" int hero_cast(const void* src, void* dst, size_t n, int sd, int dd, hero_stream_t stream) {
  HERO_REQUIRE(src && dst, "hero_cast: null pointer");
  if (n == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for(n >> 2);
  if (sd == HERO_F32 && dd == HERO_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
  else if (sd == HERO_BF16 && dd == HERO_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
  else if (sd == HERO_F32 && dd == HERO_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else if (sd == HERO_BF16 && dd == HERO_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
  else { set_error("hero_cast: bad dtypes %d -> %d", sd, dd); return HERO_ERR_ARG; }
  return check_launch("hero_cast");
}

extern "C" int hero_transpose_cast(const float* src, void* dst, int rows, int cols, int ldd, int dst_dtype, hero_stream_t stream) {
  HERO_REQUIRE(src && dst, "hero_transpose_cast: null pointer");
  if (rows <= 0 || cols <= 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  if (dst_dtype == HERO_BF16) hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, dim3(256), 0, s, src, (bf16_t*)dst, rows, cols, ldd);
  else if (dst_dtype == HERO_F32) hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, dim3(256), 0, s, src, (float*)dst, rows, cols, ldd);
  else { set_error("hero_transpose_cast: bad dtype %d", dst_dtype); return HERO_ERR_ARG; }
  return check_launch("hero_transpose_cast");
}

extern "C" int hero_copy_multi(const HeroCopyDesc* descs, const int32_t* tile_desc, const int32_t* tile_index, int n_tiles,
                               hero_stream_t stream) {
  HERO_REQUIRE(descs && tile_desc && tile_index, "hero_copy_multi: null pointer");
  if (n_tiles <= 0) return HERO_OK;
  hipLaunchKernelGGL(copy_multi_kernel, dim3(n_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), descs, tile_desc, tile_index);
  return check_launch("hero_copy_multi");
}

extern "C" int hero_relu_bwd(const void* dy, const void* y, void* dx, size_t n, int dtype, hero_stream_t stream) {
  HERO_REQUIRE(dy && y && dx, "hero_relu_bwd: null pointer");
  HERO_REQUIRE(n % 4 == 0, "hero_relu_bwd: n must be a multiple of 4");
  if (n == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for(n >> 2);
  if (dtype == HERO_F32) hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (const float*)y, (float*)dx, n);
  else if (dtype == HERO_BF16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, n);
  else { set_error("hero_relu_bwd: bad dtype %d", dtype); return HERO_ERR_ARG; }
  return check_launch("hero_relu_bwd");
}

extern "C" int hero_gelu_bwd(const void* dy, const void* u, void* dx, size_t n, int dtype, hero_stream_t stream) {
  HERO_REQUIRE(dy && u && dx, "hero_gelu_bwd: null pointer");
  HERO_REQUIRE(n % 4 == 0, "hero_gelu_bwd: n must be a multiple of 4");
  if (n == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for(n >> 2);
  if (dtype == HERO_F32) hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (const float*)u, (float*)dx, n);
  else if (dtype == HERO_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)dx, n);
  else { set_error("hero_gelu_bwd: bad dtype %d", dtype); return HERO_ERR_ARG; }
  return check_launch("hero_gelu_bwd");
}

extern "C" int hero_add(const void* a, const void* b, void* y, size_t n, int dtype, hero_stream_t stream) {
  HERO_REQUIRE(a && b && y, "hero_add: null pointer");
  HERO_REQUIRE(n % 4 == 0, "hero_add: n must be a multiple of 4");
  if (n == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for(n >> 2);
  if (dtype == HERO_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)y, n);
  else if (dtype == HERO_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n);
  else { set_error("hero_add: bad dtype %d", dtype); return HERO_ERR_ARG; }
  return check_launch("hero_add");
}

template <typename TO>
__global__ void __launch_bounds__(256) fold_slabs_kernel(const float* __restrict__ slabs, int n_slabs, size_t stride, TO* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = *reinterpret_cast<const float4*>(slabs + 4 * i);
    for (int s = 1; s < n_slabs; ++s) {                        // slab order = summation order: bit-reproducible
      const float4 b = *reinterpret_cast<const float4*>(slabs + (size_t)s * stride + 4 * i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    V4<TO>::st(out + 4 * i, a);
  }
}

extern "C" int hero_fold_slabs(const float* slabs, int n_slabs, size_t stride, void* out, size_t n, int out_dtype, hero_stream_t stream) {
  HERO_REQUIRE(slabs && out && n_slabs >= 1, "hero_fold_slabs: null pointer / no slab");
  HERO_REQUIRE(n % 4 == 0 && stride % 4 == 0 && (((uintptr_t)slabs | (uintptr_t)out) & 7) == 0, "hero_fold_slabs: n and stride must be multiples of 4, pointers aligned");
  if (n == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_for(n >> 2);
  if (out_dtype == HERO_F32) hipLaunchKernelGGL(fold_slabs_kernel<float>, dim3(grid), dim3(256), 0, s, slabs, n_slabs, stride, (float*)out, n >> 2);
  else if (out_dtype == HERO_BF16) hipLaunchKernelGGL(fold_slabs_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, slabs, n_slabs, stride, (bf16_t*)out, n >> 2);
  else { set_error("hero_fold_slabs: bad dtype %d", out_dtype); return HERO_ERR_ARG; }
  return check_launch("hero_fold_slabs");
}

extern "C" int hero_sumsq(const float* g, size_t n, float* sumsq, float* workspace, hero_stream_t stream) {
  HERO_REQUIRE(g && sumsq && workspace, "hero_sumsq: null pointer");
  if (n == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = grid_for(n >> 2);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, s, g, n, workspace);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, workspace, nb, sumsq);
  return check_launch("hero_sumsq");
}

extern "C" int hero_adamw(const HeroAdamW* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->p && a->g && a->m && a->v, "hero_adamw: null pointer");
  HERO_REQUIRE(a->step >= 1, "hero_adamw: step must be >= 1");
  if (a->n == 0) return HERO_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float bc1 = 1.f - powf(a->beta1, (float)a->step);
  const float bc2 = 1.f - powf(a->beta2, (float)a->step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(a->n >> 2)), dim3(256), 0, s, *a, bc1, bc2);
  return check_launch("hero_adamw");
}

extern "C" int hero_adamw_multi(const HeroAdamWMulti* a, hero_stream_t stream) {
  HERO_REQUIRE(a && a->descs && a->chunk_tensor && a->chunk_index, "hero_adamw_multi: null pointer");
  HERO_REQUIRE(a->step >= 1 || a->step_ptr || a->tensor_steps, "hero_adamw_multi: step must be >= 1");
  if (a->n_chunks <= 0) return HERO_OK;
  if (a->tensor_steps) {
    HERO_REQUIRE(a->n_tensors > 0, "hero_adamw_multi: tensor_steps needs n_tensors");
    hipLaunchKernelGGL(adamw_steps_inc_kernel, dim3((a->n_tensors + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a->descs,
                       a->n_tensors, a->tensor_steps);
    const int rc = check_launch("hero_adamw_multi(steps)");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(a->n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("hero_adamw_multi");
}
extern "C" int hero_adamw_multi_chunk(void) { return MT_CHUNK; }
