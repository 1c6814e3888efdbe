cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "dropout_train_mode" 2>&1 | tail -40) > $OUT/p1.log 2>&1
grep -n "Mismatch\|Greatest\|passed\|failed\|differ" $OUT/p1.log
(timeout 600 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -5) > $OUT/p2.log 2>&1; tail -3 $OUT/p2.log
HERO_NOGROUP=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -3
