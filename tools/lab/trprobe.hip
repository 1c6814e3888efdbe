// Probe the lane mapping of ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = element index.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = l * 8;                                   // consecutive 8-byte groups
  else if (mode == 1) addr = (l & 15) * 512 + (l >> 4) * 8;      // 16 rows of 256 elems (512 B), col group by l>>4
  else addr = ((l & 15) >> 2) * 512 + (l & 3) * 8 + (l >> 4) * 32; // 4 rows x 16 cols block per 16 lanes, row stride 512 B
  addr += (unsigned)(uintptr_t)lds;   // LDS base offset (address space 3 pointer value)
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[l * 4 + 0] = r.x & 0xffff; out[l * 4 + 1] = r.x >> 16; out[l * 4 + 2] = r.y & 0xffff; out[l * 4 + 3] = r.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (%s)\n", mode, hipGetErrorString(hipGetLastError()));
    for (int l = 0; l < 64; ++l) { printf("L%02d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf(l % 4 == 3 ? "\n" : "   "); }
  }
  return 0;
}
