cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 1500 python -m pytest tests/test_gpu_configs.py -q -s 2>&1 | grep -v "^tensor\|^E  " | tail -150) > $OUT/cfg_tests.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -5) > $OUT/k_tests.log 2>&1
tail -5 $OUT/k_tests.log; cat $OUT/cfg_tests.log
