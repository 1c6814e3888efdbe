#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== A: round-3 behaviour (atomic scatter, saved pre-activation)"; HERO_ATOMIC_SCATTER=1 HERO_GELU_SAVE_U=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-330
echo "== B: default"; timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-330
echo "== A again"; HERO_ATOMIC_SCATTER=1 HERO_GELU_SAVE_U=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-330
echo "== B again"; timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-330
echo "== C: atomic scatter only"; HERO_ATOMIC_SCATTER=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-330
echo "== copy sources ragged"; timeout 300 python tools/lab/copy_sources.py ragged 2>&1 | tail -30
echo "== repro tests"; timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -s -k "reproducible or long_run" 2>&1 | tail -12
} > gpurun_out/r4_run6.log 2>&1
tail -100 gpurun_out/r4_run6.log
