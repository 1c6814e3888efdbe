#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/c2_tests.log 2>&1
(timeout 600 python bench.py) > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err
(timeout 200 python tools/lab/copy_sources.py D4 2>&1 | tail -25) > gpurun_out/c2_copies_D4.txt
tail -8 gpurun_out/c2_tests.log; cut -c1-400 gpurun_out/c2_bench.json; python -c "
import json; d=json.load(open('gpurun_out/c2_bench.json')); print(d['box']); print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d['secondary'].items() if isinstance(v, dict)}); print(d['roofline']['avg_launch_us'], d['roofline']['frac'])"
cat gpurun_out/c2_copies_D4.txt
