#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -40) > gpurun_out/c4_tests.log 2>&1
bash tools/lab/r5_ab.sh > /dev/null 2>&1
(timeout 600 python bench.py) > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err
tail -12 gpurun_out/c4_tests.log; cat gpurun_out/r5_ab.log; python -c "
import json; d=json.load(open('gpurun_out/c4_bench.json')); print(d['value'], d['ms_per_step'], d['box']); print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d['secondary'].items() if isinstance(v, dict)}); print(d['roofline']['avg_launch_us'], d['roofline']['frac'])"
