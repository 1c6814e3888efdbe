# one gpurun call: the committed evidence of the round (profiles/r02_*)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
bash tools/profile_run.sh r02 > $OUT/r02_run.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 4) > $OUT/r02_bench.json 2> $OUT/r02_bench.err
for w in D2r D3; do (timeout 600 python bench.py --workload $w --steps 20 --warmup 4 2>/dev/null | grep "^{" | tail -1) > $OUT/r02_bench_$w.json; done
(timeout 1500 python bench.py --workload D4 --steps 4 --warmup 2 2>/dev/null | grep "^{" | tail -1) > $OUT/r02_bench_D4.json
tail -3 $OUT/r02_run.log; cut -c1-200 $OUT/r02_bench.json; for w in D2r D3 D4; do cut -c1-400 $OUT/r02_bench_$w.json | tail -c 260; echo; done
