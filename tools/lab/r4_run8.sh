#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_head.py tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_pretrain.py tests/test_gpu_collate.py -q -x 2>&1 | tail -8
echo "== repro tests"; timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -s -k "reproducible or long_run" 2>&1 | tail -12
echo "== repro again"; timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -s -k "reproducible" 2>&1 | tail -6
echo "== bench D2"; timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-330
} > gpurun_out/r4_run8.log 2>&1
tail -60 gpurun_out/r4_run8.log
