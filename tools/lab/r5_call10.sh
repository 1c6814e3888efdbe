#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -8) > gpurun_out/c10_tests.log 2>&1
bash tools/lab/r5_ab.sh > /dev/null 2>&1
cat gpurun_out/c10_tests.log gpurun_out/r5_ab.log
