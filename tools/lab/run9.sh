cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_collate.py tests/test_gpu_distributed.py -q 2>&1 | tail -30) > $OUT/collate_tests.log 2>&1
tail -12 $OUT/collate_tests.log
(timeout 600 python bench.py --workload D3 --steps 10 --warmup 2 2>&1 | tail -12) > $OUT/bench_D3.log 2>&1; tail -4 $OUT/bench_D3.log
(timeout 1500 python bench.py --workload D4 --steps 2 --warmup 0 2>&1 | tail -6) > $OUT/bench_D4.log 2>&1; tail -3 $OUT/bench_D4.log
