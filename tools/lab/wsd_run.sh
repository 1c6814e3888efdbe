#!/bin/bash
# round 4: deferred-epilogue GEMM - correctness, A/B micro-benchmark, A/B of the bench step
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wave_specialised" 2>&1 | tail -15
echo "== wsd_ab"; timeout 300 python tools/lab/wsd_ab.py 2>&1 | tail -80
echo "== bench inline"; HERO_WS_DEFER=0 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -3
echo "== bench deferred"; HERO_WS_DEFER=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -3
} > gpurun_out/wsd_run.log 2>&1
tail -120 gpurun_out/wsd_run.log
