cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15) > $OUT/all_gpu_tests.log 2>&1
tail -8 $OUT/all_gpu_tests.log
