#!/bin/bash
# Same-box A/B of the 64-row wave-specialised geometries for the small-M GEMMs (HERO_WS_SMALL_M=0: the 4-wave kernels).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { echo "== $1"; env $2 timeout 300 python bench.py --no-cpu-baseline --no-secondary $3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
{
run default X=1
run 4wave HERO_WS_SMALL_M=0
run default X=1
run 4wave HERO_WS_SMALL_M=0
run default_D2r X=1 "--workload D2r"
run 4wave_D2r HERO_WS_SMALL_M=0 "--workload D2r"
run default_D3 X=1 "--workload D3"
run 4wave_D3 HERO_WS_SMALL_M=0 "--workload D3"
} > gpurun_out/ab_small_m.log 2>&1
cat gpurun_out/ab_small_m.log
