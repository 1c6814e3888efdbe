// Helpers shared by the matrix-core attention kernels (attention_mfma.hip: L <= 64, one wave per head;
// attention_mfma_long.hip: 64 < L <= 256, one workgroup per head).  bf16, head size 64.
#pragma once
#include "common.h"

namespace hero {
namespace attn {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

constexpr int RS = 72;   // LDS row stride (bf16 elements) of a [rows][64] head tile: 144 B, conflict-free b128 / tr reads

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// [L][64] bf16 head slice (row stride ld) -> wave-private LDS tile [32*NB][RS], rows >= L zeroed
template <int NB>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, int ld, int L, bf16_t* dst, int lane) {
  uint4 v[4 * NB];
  const int c = (lane & 7) * 8;
#pragma unroll
  for (int it = 0; it < 4 * NB; ++it) {
    const int r = it * 8 + (lane >> 3);
    v[it] = *reinterpret_cast<const uint4*>(src + (size_t)min(r, L - 1) * ld + c);
  }
#pragma unroll
  for (int it = 0; it < 4 * NB; ++it) {
    const int r = it * 8 + (lane >> 3);
    uint4 t = v[it];
    if (r >= L) t = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(dst + r * RS + c) = t;
  }
}

// row fragment (MFMA A or B operand with the contraction along the row): 8 consecutive elements
__device__ __forceinline__ bf16x8_t gfrag(const bf16_t* __restrict__ base, int ld, int row, int L, int ks, int half) {
  return *reinterpret_cast<const bf16x8_t*>(base + (size_t)min(row, L - 1) * ld + 16 * ks + 8 * half);
}
__device__ __forceinline__ bf16x8_t lfrag(const bf16_t* tile, int row, int ks, int half) {
  return *reinterpret_cast<const bf16x8_t*>(tile + row * RS + 16 * ks + 8 * half);
}

// transposed fragment: the lane's 8 k-slots are rows (ra .. ra+3) and (rb .. rb+3) of an LDS tile,
// its m/n index is column cb*32 + (lane & 31).  `stride_b` = row stride in bytes.
__device__ __forceinline__ unsigned tr_addr(const void* tile, int stride_b, int row0, int cb, int lane) {
  const int p = lane & 15, gq = (lane >> 4) & 1;
  return (unsigned)(uintptr_t)tile + (row0 + (p >> 2)) * stride_b + (cb * 32 + gq * 16 + 4 * (p & 3)) * 2;
}
__device__ __forceinline__ bf16x8_t tr_frag(unsigned addr_a, unsigned addr_b) {
  uint2 r0, r1;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %2\n\t"
      "ds_read_b64_tr_b16 %1, %3\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r0), "=&v"(r1)
      : "v"(addr_a), "v"(addr_b)
      : "memory");
  const u32x4_t t = {r0.x, r0.y, r1.x, r1.y};
  return __builtin_bit_cast(bf16x8_t, t);
}

__device__ __forceinline__ bf16x8_t pack8(const float* v) {
  const u32x4_t t = {f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7])};
  return __builtin_bit_cast(bf16x8_t, t);
}
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// key (or, in the backward's LDS round trip, query) index of accumulator register r in a 32x32 tile
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// 8-byte store of 4 consecutive head dims
__device__ __forceinline__ void st_bf4(bf16_t* p, float a, float b, float c, float d) {
  uint2 u;
  u.x = f2bf_pk(a, b);
  u.y = f2bf_pk(c, d);
  *reinterpret_cast<uint2*>(p) = u;
}


// cooperative variant of stage_tile: NT threads copy rows [0, R) of a [L][64] head slice, rows >= L zeroed
template <int NT>
__device__ __forceinline__ void stage_tile_wg(const bf16_t* __restrict__ src, int ld, int L, int R, bf16_t* dst) {
  for (int q = threadIdx.x; q < R * 8; q += NT) {
    const int r = q >> 3, c = (q & 7) * 8;
    uint4 t = *reinterpret_cast<const uint4*>(src + (size_t)min(r, L - 1) * ld + c);
    if (r >= L) t = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(dst + r * RS + c) = t;
  }
}

// one 32-row tile of out^T[dt] (head dim x lane-owned row) -> out[row][d], rows < L
__device__ __forceinline__ void store_tileT(bf16_t* __restrict__ dst, int ld, int L, int tile, const f32x16_t (&acc)[2], int lane) {
  const int half = lane >> 5, row = 32 * tile + (lane & 31);
  if (row < L) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        st_bf4(dst + (size_t)row * ld + 32 * dt + 8 * q + 4 * half, acc[dt][4 * q], acc[dt][4 * q + 1], acc[dt][4 * q + 2],
               acc[dt][4 * q + 3]);
  }
}

}  // namespace attn
}  // namespace hero
