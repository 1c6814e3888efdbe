#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats + the two PMC passes for the
# bench step (eager launches so that every kernel is a separate dispatch), summaries into gpurun_out/.
#   tools/profile_run.sh <tag>      e.g. r01b
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-graph --no-box-probe > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-graph --no-box-probe --profile-steps 0 > /dev/null 2> $OUT/${TAG}_pmcf.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-graph --no-box-probe --profile-steps 0 > /dev/null 2> $OUT/${TAG}_pmcw.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma -o m -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-graph --no-box-probe --profile-steps 0 > /dev/null 2> $OUT/${TAG}_pmcm.log
cd $ROOT
python tools/profile_summary.py mfma $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_mfma.json
# steady-state micro-steps only (the trace holds 2 prepare + 2 warm-up + 6 timed + 2 roofline-leg micro-steps = 7 optimiser steps; the first 2 are skipped)
python tools/profile_summary.py steady $OUT/${TAG}_stats 2 $OUT/${TAG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-graph --no-box-probe --no-box-probe (MI355X)"
python tools/profile_summary.py pmc $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_traffic.json
# raw per-dispatch traces are large; keep only the summaries + stats csv
rm -f $OUT/${TAG}_pmc_fetch/*/*counter_collection.csv $OUT/${TAG}_pmc_write/*/*counter_collection.csv 2>/dev/null
find $OUT/${TAG}_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma -name "*counter_collection.csv" -delete 2>/dev/null
head -30 $OUT/${TAG}_kernel_stats.csv
