#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'))"; }
{
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_configs.py tests/test_gpu_collate.py tests/test_gpu_loader.py tests/test_gpu_pretrain.py -q -x 2>&1 | tail -4
echo "== per-group attention"; HERO_ATTN_PER_GROUP=1 timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | val
echo "== one launch"; timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | val
echo "== per-group attention"; HERO_ATTN_PER_GROUP=1 timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | val
echo "== one launch"; timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | val
} > gpurun_out/r4_run14.log 2>&1
cat gpurun_out/r4_run14.log
