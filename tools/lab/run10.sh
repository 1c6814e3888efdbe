cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_distributed.py -x -q -k "gemm or two_rank or bench_two" 2>&1 | tail -8) > $OUT/k_tests.log 2>&1
tail -5 $OUT/k_tests.log
(timeout 300 python tools/gemm_bench.py wgrad -1; timeout 300 python tools/gemm_bench.py wgrad 8) > $OUT/wgrad_bench.log 2>&1; cat $OUT/wgrad_bench.log
(timeout 300 python bench.py --steps 20 --no-cpu-baseline) > $OUT/ws_step.log 2>&1; tail -1 $OUT/ws_step.log | cut -c1-400
