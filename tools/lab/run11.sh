cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or wgrad or deferred" 2>&1 | tail -12) > $OUT/k_tests.log 2>&1
tail -6 $OUT/k_tests.log
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -8) > $OUT/p_tests.log 2>&1
tail -5 $OUT/p_tests.log
(timeout 300 python bench.py --steps 20 --no-cpu-baseline) > $OUT/ws_step.log 2>&1; tail -1 $OUT/ws_step.log | cut -c1-330
(timeout 300 python tools/torch_ops_sources.py 2>&1 | tail -45) > $OUT/torch_ops.log 2>&1; cat $OUT/torch_ops.log
