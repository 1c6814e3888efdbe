cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29561 tools/lab/dbg_worker.py > gpurun_out/dbg.log 2>&1
grep -n "PENDING\|LAUNCH" gpurun_out/dbg.log | head -12
