#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_loader.py -x -q -k "same_destination or meta_loader or deferred" 2>&1 | tail -5
echo "== fill sources"; timeout 300 python tools/torch_fill_sources.py 2>&1 | tail -70
echo "== census"; timeout 300 python tools/gemm_census.py 2>&1 | tail -60
} > gpurun_out/r4_run2.log 2>&1
tail -150 gpurun_out/r4_run2.log
