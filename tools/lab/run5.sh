cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or gemm" 2>&1 | tail -15) > $OUT/k_tests.log 2>&1
(timeout 1500 python -m pytest tests/test_gpu_configs.py -q -s -k "d2_batch or long_video" 2>&1 | grep -v "^tensor\|^E  " | tail -120) > $OUT/cfg_tests.log 2>&1
(timeout 300 python tools/gemm_bench.py wgrad -1; timeout 300 python tools/gemm_bench.py wgrad 8) > $OUT/wsk_bench.log 2>&1
tail -6 $OUT/k_tests.log; cat $OUT/cfg_tests.log; cat $OUT/wsk_bench.log
