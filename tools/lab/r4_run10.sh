#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { echo "== $1"; env $2 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"; }
{
echo "== r3 tree"; (cd .ab/9320138 && timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])")
run default X=1
run b1_epilogue HERO_B1_EPILOGUE=1
run atomic_scatter HERO_ATOMIC_SCATTER=1
run both "HERO_B1_EPILOGUE=1 HERO_ATOMIC_SCATTER=1"
run both+save_u "HERO_B1_EPILOGUE=1 HERO_ATOMIC_SCATTER=1 HERO_GELU_SAVE_U=1"
run default X=1
echo "== r3 tree"; (cd .ab/9320138 && timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])")
} > gpurun_out/r4_run10.log 2>&1
cat gpurun_out/r4_run10.log
