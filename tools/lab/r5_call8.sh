#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -30) > gpurun_out/c8_tests.log 2>&1
bash tools/lab/r5_ab.sh > /dev/null 2>&1
(timeout 200 python tools/torch_fill_sources.py 2>&1 | tail -24) > gpurun_out/c8_fills.txt
tail -8 gpurun_out/c8_tests.log; cat gpurun_out/r5_ab.log; cat gpurun_out/c8_fills.txt
