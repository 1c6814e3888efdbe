cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d2r_now -o s -- python $R/bench.py --workload D2r --steps 4 --warmup 2 --no-graph > /dev/null 2> $R/gpurun_out/d2r_now.log)
python tools/profile_summary.py stats gpurun_out/d2r_now 8 gpurun_out/d2r_now_stats.csv "D2r now"
find gpurun_out/d2r_now -name "*kernel_trace.csv" -delete
grep attn gpurun_out/d2r_now_stats.csv
