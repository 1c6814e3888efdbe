# one gpurun call: the committed evidence of the round (profiles/r0N_*): kernel stats + PMC passes of the bench step,
# kernel stats of the secondary workloads, the bench lines.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r06}
bash tools/profile_run.sh $TAG > $OUT/${TAG}_run.log 2>&1
export TMPDIR=/tmp
R=$PWD
for w in D2r D3; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${TAG}_stats_$w -o s -- python $R/bench.py --workload $w --steps 4 --warmup 2 --no-graph > /dev/null 2> $R/$OUT/${TAG}_stats_$w.log)
  # micro-steps in the trace: D2r 2 (prepare) + 2 (warm-up) + 4; D3: 4 tasks x (2 + 2 + 4)
  python tools/profile_summary.py steady $OUT/${TAG}_stats_$w 1 $OUT/${TAG}_kernel_stats_$w.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w --steps 4 --warmup 2 --no-graph (MI355X)"
  find $OUT/${TAG}_stats_$w -name "*kernel_trace.csv" -delete
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${TAG}_stats_D4 -o s -- python $R/bench.py --workload D4 --videos 256 --steps 4 --warmup 2 > /dev/null 2> $R/$OUT/${TAG}_stats_D4.log)
python tools/profile_summary.py steady $OUT/${TAG}_stats_D4 1 $OUT/${TAG}_kernel_stats_D4.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload D4 --videos 256 --steps 4 --warmup 2 (MI355X; 256 videos x 256 frames per step)"
find $OUT/${TAG}_stats_D4 -name "*kernel_trace.csv" -delete
# idle time between the kernels of the hipGraph replay (tools/lab/graph_gaps.py)
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/${TAG}_gaps -o g -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --profile-steps 0 > /dev/null 2> $R/$OUT/${TAG}_gaps.log)
python tools/lab/graph_gaps.py $OUT/${TAG}_gaps > $OUT/${TAG}_graph_gaps.txt 2>&1
find $OUT/${TAG}_gaps -name "*kernel_trace.csv" -delete
# the line reads its counters from profiles/ (stamped with the kernel-source hash): the passes of THIS call
cp $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_mfma.json profiles/
(timeout 600 python bench.py --steps 20 --warmup 4) > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
(timeout 600 python bench.py --steps 20 --warmup 4 --feed 4 --no-cpu-baseline | tail -1) > $OUT/${TAG}_bench_feed.json 2>> $OUT/${TAG}_bench.err
for w in D2r D3 D4; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline 2> $OUT/${TAG}_$w.err | tail -1 > $OUT/${TAG}_bench_$w.json
done
timeout 300 python bench.py --workload D2r --no-graph --no-cpu-baseline 2>> $OUT/${TAG}_D2r.err | tail -1 > $OUT/${TAG}_bench_D2r_eager.json
(timeout 300 python tools/lab/feedprobe.py 2>&1 | grep -v Warning | tail -22) > $OUT/${TAG}_feedprobe.txt
tail -3 $OUT/${TAG}_run.log; cat $OUT/${TAG}_feedprobe.txt; cut -c1-200 $OUT/${TAG}_bench.json; cut -c1-200 $OUT/${TAG}_bench_feed.json
for w in D2r D3 D4; do head -8 $OUT/${TAG}_kernel_stats_$w.csv | cut -c1-140; done
