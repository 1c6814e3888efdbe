cd $GRAFT_REPO_ROOT
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -15) > $OUT/attn_tests.log 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02a_stats -o s -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/r02a_bench_under_rocprof.json 2> $OUT/r02a_stats.log
cd $ROOT
python tools/profile_summary.py stats $OUT/r02a_stats 12 $OUT/r02a_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph (MI355X)"
find $OUT/r02a_stats -name "*kernel_trace.csv" -delete 2>/dev/null
tail -5 $OUT/attn_tests.log; head -60 $OUT/r02a_kernel_stats.csv
