"""2-rank data parallelism with the real HIP kernels on the one GPU of the test box (see
tests/dist_worker.py for what is checked)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("wire", ["none", "bf16"])
def test_two_rank_data_parallel_real_kernels(tmp_path, wire):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), wire]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert res[0]["buckets"] > 3                      # several overlapped buckets were exercised
    assert res[0]["losses"] != res[1]["losses"]       # different data per rank, same weights


def test_bench_two_ranks_end_to_end_on_one_device():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), with
    both ranks on the box's single GPU and gloo as the transport (HERO_BENCH_ONE_DEVICE / HERO_BENCH_BACKEND):
    process-group setup, parameter broadcast, bucketed bf16 gradient all-reduce overlapped with backward,
    cross-rank negatives, max-over-ranks timing and the one JSON line on rank 0."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HERO_BENCH_ONE_DEVICE="1",
               HERO_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "4", "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 64 and out["config"]["launch"].startswith(("eager", "hipGraph replay (step captured"))
    assert out["final_loss"] == out["final_loss"] and out["roofline"]["frac"] > 0      # finite loss, GEMM events recorded
    # the line is self-describing: how many ranks the collective saw, what the exchange costs alone, what of it is exposed
    c = out["comm"]
    assert c["ranks_seen"] == 2 and c["backend"].startswith("gloo") and c["wire_dtype"] == "bf16" and c["buckets"] > 3
    assert c["allreduce_ms_per_opt_step"] > 0 and c["eager_ms_per_step"] > 0 and c["eager_ms_per_step_without_grad_exchange"] > 0
    assert c["exposed_ms_per_opt_step"] >= 0 and 200 < c["payload_mb_per_opt_step"] < 260          # 121 M parameters x 2 B


@pytest.mark.parametrize("wire", ["none", "bf16"])
def test_one_rank_over_rccl(tmp_path, wire):
    """The real RCCL backend (`init_process_group("nccl", device_id=...)`) with one rank and
    HERO_DP_FORCE_COLLECTIVES=1: every gradient bucket goes through an asynchronous RCCL all-reduce issued from the
    backward hooks (fp32 in place / bf16 wire buffers), finish() waits on RCCL's stream, the optimiser follows."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HERO_DP_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), wire, "nccl"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(tmp_path / "rank0.json"))
    assert res["backend"] == "nccl" and res["collectives"] and res["buckets"] > 3


def test_bench_one_rank_over_rccl():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, eager launches, RCCL process group, bucketed
    bf16 all-reduce overlapped with backward), with the one rank a 1-GPU box can host."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HERO_DP_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29549", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["launch"].startswith(("eager", "hipGraph replay (step captured"))
    assert out["final_loss"] == out["final_loss"]
    print("bench over RCCL (1 rank, forced collectives):", out["value"], "videos/s", out["ms_per_step"], "ms", out["config"])


def test_bench_one_rank_over_rccl_captured_in_a_hipgraph():
    """The data-parallel bench first times the eager step, then captures the step WITH its collectives - the bucketed
    RCCL all-reduces issued from the backward hooks (bf16 wire buffers) and finish()'s waits become nodes of the boundary
    graph - and times the replay; the line reports the faster run and keeps the other one's time (eager N > 1 is
    host-bound once the kernels are fast: ~5.5 ms of host issue vs ~6.9 ms of GPU work per micro-step).  One rank is
    what a 1-GPU box can host.  HERO_DP_GRAPH=0 keeps the run eager."""
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HERO_DP_FORCE_COLLECTIVES="1")
    res = {}
    for mode, port in (("1", "29551"), ("0", "29553")):
        env = dict(base, HERO_DP_GRAPH=mode)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
               "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    g, e = res["1"], res["0"]
    assert e["config"]["launch"] == "eager" and "graph_ms_per_step" not in e["config"]
    assert g["final_loss"] == g["final_loss"]
    print("RCCL one rank:", g["config"]["launch"], g["ms_per_step"], {k: v for k, v in g["config"].items() if k.endswith("_ms_per_step")})
    assert g["config"]["launch"].startswith("hipGraph replay (step captured") and g["config"]["eager_ms_per_step"] >= g["ms_per_step"]


def test_bench_wedged_capture_falls_back_to_the_eager_line():
    """the captured-collectives run is a second, optional measurement - if it never comes back
    (HERO_DP_GRAPH_TEST_WEDGE simulates a wedged collective) the watchdog prints the eager line, which already carries
    the `comm` block, and every rank leaves with exit code 0 so that the launcher does not hang."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HERO_DP_FORCE_COLLECTIVES="1",
               HERO_DP_GRAPH_TEST_WEDGE="1", HERO_DP_GRAPH_TIMEOUT="5")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29557", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "timed out" in out["config"]["launch"] and out["config"]["launch"].startswith("eager")
    assert out["value"] > 0 and out["comm"]["ranks_seen"] == 1 and out["comm"]["graph_ms_per_step"] is None
