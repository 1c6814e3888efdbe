"""2-rank data parallelism with the real HIP kernels on the one GPU of the test box (see
tests/dist_worker.py for what is checked)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_data_parallel_real_kernels(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert res[0]["buckets"] > 3                      # several overlapped buckets were exercised
    assert res[0]["losses"] != res[1]["losses"]       # different data per rank, same weights
