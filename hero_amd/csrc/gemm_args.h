// Types shared by the GEMM translation units (gemm.hip: 4-wave tiles, gemm_ws.hip: wave-specialised
// persistent tiles).
#pragma once
#include "common.h"

namespace hero {

// compile-time epilogue selection of the hot-path combinations (everything else is EK_GENERIC)
enum { EK_GENERIC = 0x100, EK_BIAS = 1, EK_GELU = 2, EK_RES = 4, EK_DROP = 8, EK_GELU_BWD = 16 };

// per-launch timing with HIP events on the launch stream (bench.py roofline leg; gemm.hip owns the slots)
void* gemm_prof_begin(int slot, hipStream_t s);
void gemm_prof_end(void* token, double flops, hipStream_t s);

int gemm_forced_config();     // hero_gemm_force_config state (gemm.hip): -1 heuristic, 8 = 4-wave kernels only, 9 = wave-specialised always

// gemm_ws.hip.  Returns -1 when the problem is outside this kernel family (caller falls through to the
// 4-wave kernels), otherwise the launch status.  C[M,N] (+)= op(A) op(B); see hero_gemm for the layouts.
int gemm_ws_run(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int a_layout,
                int b_layout, const HeroGemmEpilogue& epi, int force_cfg, hipStream_t s);

}  // namespace hero
