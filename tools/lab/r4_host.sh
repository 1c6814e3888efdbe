#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== host profile"; timeout 300 python tools/host_profile.py 2>&1 | tail -12
echo "== bench, 1 rank over RCCL (forced collectives)"; HERO_DP_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | grep "^{" | tail -1 > gpurun_out/r04_bench_rccl_1rank.json; cut -c1-200 gpurun_out/r04_bench_rccl_1rank.json; python -c "import json; d=json.load(open('gpurun_out/r04_bench_rccl_1rank.json')); print(json.dumps(d['comm'], indent=1)); print(d['config']['launch'], d['ms_per_step'])"
} > gpurun_out/r4_host.log 2>&1
cat gpurun_out/r4_host.log
