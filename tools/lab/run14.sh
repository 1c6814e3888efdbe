cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_distributed.py -q -x -k "two_rank and none" 2>&1 | grep -v "^E   *$" | tail -60) > $OUT/dist1.log 2>&1
grep -n "Error\|error\|assert\|RANK\|Traceback" $OUT/dist1.log | head -30
(timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -k "graph_replay" 2>&1 | tail -30) > $OUT/long1.log 2>&1
grep -n "assert\|Error\|tail\|tensor" $OUT/long1.log | head -20
