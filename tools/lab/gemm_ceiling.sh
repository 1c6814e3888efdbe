#!/bin/bash
# profiles/r05_gemm_ceiling.txt: product K,K kernels vs their main loops alone vs the vendor library, per
# launch of the step, against the box's measured MFMA / HBM peaks.  Run on the GPU box from the repo root; the noepi build
# must exist (tools/lab/build_variants.sh noepi:-DHERO_WS_NOEPI, done in the build container - hipcc is slow on the box).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 300 python tools/lab/gemm_ceiling.py product gpurun_out/ceil_product.json > gpurun_out/ceil_product.log 2>&1
HERO_HIP_LIB=$PWD/tools/lab/libhero_noepi.so timeout 300 python tools/lab/gemm_ceiling.py noepi gpurun_out/ceil_noepi.json > gpurun_out/ceil_noepi.log 2>&1
python tools/lab/gemm_ceiling.py table gpurun_out/ceil_product.json gpurun_out/ceil_noepi.json > gpurun_out/r05_gemm_ceiling.txt 2>&1
cat gpurun_out/r05_gemm_ceiling.txt
