#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('peak_mem_gb'), (d.get('roofline') or {}).get('avg_launch_us'), d['config'].get('workload','')[:40])"; }
{
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_head.py tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_configs.py tests/test_gpu_pretrain.py -q -x 2>&1 | tail -4
echo "== r3 tree D2"; (cd .ab/9320138 && timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | val)
echo "== now D2"; timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | val
echo "== r3 tree D2"; (cd .ab/9320138 && timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | val)
echo "== now D2"; timeout 400 python bench.py --no-cpu-baseline --no-secondary 2>&1 | val
echo "== D4 256 videos, flush at the soft cap (round 3)"; HERO_WGRAD_QUEUE_HARD_MB=4096 timeout 600 python bench.py --workload D4 --videos 256 --steps 4 --warmup 2 2>&1 | val
echo "== D4 256 videos, new policy"; timeout 600 python bench.py --workload D4 --videos 256 --steps 4 --warmup 2 2>&1 | val
echo "== D4 sized to HBM, new policy"; timeout 900 python bench.py --workload D4 2>&1 | val
} > gpurun_out/r4_run13.log 2>&1
cat gpurun_out/r4_run13.log
