#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_configs.py tests/test_gpu_pretrain.py -q -x 2>&1 | tail -6
echo "== bench D2"; timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-330
echo "== D4 256 videos, old policy (hard cap = soft cap)"; HERO_WGRAD_QUEUE_HARD_MB=4096 timeout 600 python bench.py --workload D4 --videos 256 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-260
echo "== D4 256 videos, new policy"; timeout 600 python bench.py --workload D4 --videos 256 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-260
echo "== D4 sized to HBM, new policy"; timeout 900 python bench.py --workload D4 2>&1 | tail -1 | cut -c1-700
} > gpurun_out/r4_run12.log 2>&1
tail -40 gpurun_out/r4_run12.log
