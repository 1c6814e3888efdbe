#!/bin/bash
# Same-box A/B of the round-4 environment switches (each restores a round-3 behaviour):
#   HERO_B1_EPILOGUE=1     FFN1 bias gradient from the gelu' epilogue's fp32 atomics
#   HERO_ATOMIC_SCATTER=1  fp32-atomic embedding scatter
#   HERO_GELU_SAVE_U=1     FFN1 saves the pre-activation
#   HERO_ATTN_PER_GROUP=1  one attention launch per sequence group
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { echo "== $1"; env $2 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"; }
{
run default X=1
run b1_epilogue HERO_B1_EPILOGUE=1
run atomic_scatter HERO_ATOMIC_SCATTER=1
run save_u HERO_GELU_SAVE_U=1
run attn_per_group HERO_ATTN_PER_GROUP=1
run round3_like "HERO_B1_EPILOGUE=1 HERO_ATOMIC_SCATTER=1 HERO_GELU_SAVE_U=1 HERO_ATTN_PER_GROUP=1"
run default X=1
} > gpurun_out/ab_switches.log 2>&1
cat gpurun_out/ab_switches.log
