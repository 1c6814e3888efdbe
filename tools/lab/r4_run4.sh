#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25
echo "== bench D2"; timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1
echo "== bench D3"; timeout 300 python bench.py --no-cpu-baseline --workload D3 2>&1 | tail -1 | cut -c1-1200
echo "== fill sources ragged"; timeout 300 python tools/torch_fill_sources.py ragged 2>&1 | tail -40
} > gpurun_out/r4_run4.log 2>&1
tail -100 gpurun_out/r4_run4.log
