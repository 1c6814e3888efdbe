#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -30) > gpurun_out/c7_tests.log 2>&1
bash tools/lab/final_profiles.sh r05 > gpurun_out/c7_final.log 2>&1
tail -8 gpurun_out/c7_tests.log; head -30 gpurun_out/c7_final.log | cut -c1-220
