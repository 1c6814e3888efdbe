#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== repro tests"; timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -s -k "reproducible or long_run or full_d2" 2>&1 | tail -40
echo "== copy sources ragged"; timeout 300 python tools/lab/copy_sources.py ragged 2>&1 | tail -30
} > gpurun_out/r4_run7.log 2>&1
tail -100 gpurun_out/r4_run7.log
