#!/bin/bash
# round 5, closing call: full GPU suite on the final tree, the bench line again, the N > 1 lines a 1-GPU box can produce
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu --maxfail=6 -q 2>&1 | tail -8) > gpurun_out/c9_tests.log 2>&1
(timeout 600 python bench.py) > gpurun_out/r05_bench_final.json 2> gpurun_out/c9_bench.err
(timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline | tail -1) > gpurun_out/r05_bench_gpus2_plumbing.json 2> gpurun_out/c9_g2.err
(HERO_DP_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1) > gpurun_out/r05_bench_rccl_1rank.json
(HERO_DP_FORCE_COLLECTIVES=1 MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29573 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --exchange abi 2>/dev/null | grep "^{" | tail -1) > gpurun_out/r05_bench_hero_comm_1rank.json
tail -3 gpurun_out/c9_tests.log
for f in r05_bench_final r05_bench_gpus2_plumbing r05_bench_rccl_1rank r05_bench_hero_comm_1rank; do python -c "
import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['value'], d['ms_per_step'], d['config']['parallelism'][:60], d['config']['launch'][:60]); print('   comm', {k:d['comm'][k] for k in ('exchange','ranks_seen','buckets','eager_ms_per_step','graph_ms_per_step')} if d.get('comm') else None)"; done
