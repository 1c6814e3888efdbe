#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== tests"; timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25
echo "== bench D2"; timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
echo "== bench D3"; timeout 300 python bench.py --no-cpu-baseline --workload D3 2>&1 | tail -1 | cut -c1-600
} > gpurun_out/r4_run3.log 2>&1
tail -60 gpurun_out/r4_run3.log
