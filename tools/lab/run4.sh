cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -15) > $OUT/attn_tests.log 2>&1
(timeout 1500 python -m pytest tests/test_gpu_configs.py -q 2>&1 | tail -40) > $OUT/cfg_tests.log 2>&1
tail -6 $OUT/attn_tests.log; tail -40 $OUT/cfg_tests.log
