#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -40) > gpurun_out/c3_tests.log 2>&1
bash tools/lab/r5_ab.sh > /dev/null 2>&1
(timeout 300 python tools/lab/feedprobe.py 2>&1 | grep -v Warning | tail -22) > gpurun_out/c3_feedprobe.txt
tail -12 gpurun_out/c3_tests.log; cat gpurun_out/r5_ab.log; cat gpurun_out/c3_feedprobe.txt
