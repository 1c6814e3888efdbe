cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$PWD
for v in prod nodrop; do
  if [ $v = nodrop ]; then export HERO_HIP_LIB=$R/tools/lab/libhero_nodrop.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -o s -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2> $R/gpurun_out/ab_$v.log)
  python tools/profile_summary.py stats gpurun_out/ab_$v 12 gpurun_out/ab_${v}_stats.csv "$v"
  find gpurun_out/ab_$v -name "*kernel_trace.csv" -delete
done
