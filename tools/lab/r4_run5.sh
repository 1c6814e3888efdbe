#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_loader.py tests/test_gpu_parity.py tests/test_gpu_step.py -q -x 2>&1 | tail -8
echo "== dist tests"; timeout 900 python -m pytest tests/test_gpu_distributed.py -q -x 2>&1 | tail -8
echo "== copy sources ragged"; timeout 300 python tools/lab/copy_sources.py ragged 2>&1 | tail -36
echo "== bench D2"; timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
} > gpurun_out/r4_run5.log 2>&1
tail -100 gpurun_out/r4_run5.log
