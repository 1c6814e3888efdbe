cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pretrain.py tests/test_gpu_step.py -x -q -k "cross_entropy or matmul_nt or mlm or mfm or fom or graph or first_optimizer" 2>&1 | tail -25) > $OUT/ce_tests.log 2>&1
tail -8 $OUT/ce_tests.log
for w in D2r D3; do (timeout 600 python bench.py --workload $w --steps 10 --warmup 2 2>&1 | tail -3) > $OUT/bench_$w.log 2>&1; tail -2 $OUT/bench_$w.log; done
(timeout 900 python bench.py --workload D4 --videos 64 --steps 2 --warmup 0 2>&1 | tail -3) > $OUT/bench_D4_64.log 2>&1; tail -2 $OUT/bench_D4_64.log
