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


def test_two_ranks_feed_ragged_batches_through_buckets(tmp_path):
    """Round 6: BucketedBatchFeeder under data parallelism (two ranks on the box's one GPU, gloo transport, eager launches):
    each rank pads its own ragged batches to the SHARED bucket shapes, the ranks sit on different buckets in the same step,
    gradients and cross-rank negatives are exchanged as ever, the replicas stay bit-identical through four optimiser steps."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29571",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "none", "gloo", "feeder"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert res[0]["n_buckets"] >= 2 and res[0]["losses"] != res[1]["losses"]
    assert any(a != b for a, b in zip(res[0]["buckets_used"], res[1]["buckets_used"]))      # the ranks really were on different buckets
    assert len(set(res[0]["buckets_used"]) | set(res[1]["buckets_used"])) >= 2


def test_two_ranks_several_queries_per_video_fused_head(tmp_path):
    """Round 6: the HIP head on VSM batches with three queries per video under two-rank data parallelism (cross-rank
    negatives gathered, gradients exchanged in buckets) == the PyTorch head; the global loss is the same on both ranks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "none", "gloo", "multiq"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert res[0]["nq"] == 3 * res[0]["nv"] and res[0]["rel_err"] < 2e-4 and res[1]["rel_err"] < 2e-4


def test_bench_plain_command_launches_two_ranks_on_one_device():
    """`python bench.py --gpus 2` as a PLAIN command (no launcher, no WORLD_SIZE): bench.py re-executes itself under
    torch.distributed.run (one process per rank on 127.0.0.1 - the driver's own N > 1 command line, which
    test_bench_one_rank_over_rccl covers as such); on this 1-GPU box both ranks share the device and talk over gloo, and the
    line says it is a plumbing run.  Covered end to end: process-group setup, parameter broadcast, bucketed bf16 gradient
    all-reduce overlapped with backward, cross-rank negatives, max-over-ranks timing, the one JSON line on rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HERO_BENCH_ONE_DEVICE",
                                                            "HERO_BENCH_BACKEND", "HERO_DP_FORCE_COLLECTIVES")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 64 and out["config"]["launch"].startswith(("eager", "hipGraph replay (step captured"))
    assert "PLUMBING" in out["config"]["parallelism"] and out["config"]["parallelism"].startswith("dp2")
    assert out["final_loss"] == out["final_loss"] and out["roofline"]["frac"] > 0      # finite loss, GEMM events recorded
    # the line is self-describing: how many ranks the collective saw, what the exchange costs alone, what of it is exposed
    c = out["comm"]
    assert c["ranks_seen"] == 2 and c["backend"].startswith("gloo") and c["wire_dtype"] == "bf16" and c["buckets"] > 3
    assert c["exchange"].startswith("torch.distributed")
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
    # the line reports the faster of the two measured runs and keeps the other one's time: on most boxes the replay wins
    # (eager is host-bound), on a box with a fast host the two are within noise - either way both times are on the line
    assert g["comm"]["graph_ms_per_step"] is not None and g["comm"]["graph_ms_per_step"] > 0
    if g["config"]["launch"].startswith("hipGraph replay (step captured"):
        assert g["config"]["eager_ms_per_step"] >= g["ms_per_step"]
    else:
        assert g["config"]["launch"] == "eager" and g["config"]["graph_ms_per_step"] >= g["ms_per_step"]


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


def test_hero_comm_abi_collectives_one_rank_eager_and_captured():
    """hero_comm_* of the C ABI (include/hero_hip.h) without torch.distributed: a 1-rank RCCL communicator, the bucket
    group all-reduce (fp32 + bf16 in one group), broadcast, all-gather - and the same all-reduce forked onto the
    communicator's side stream inside a hipGraph capture (no capture_error_mode, no watchdog: the collective is an ordinary
    node of the captured stream)."""
    import torch
    from hero_amd.utils import comm
    c = comm.Communicator()
    try:
        assert (c.rank, c.world) == (0, 1)
        g = torch.Generator(device="cuda").manual_seed(3)
        a = torch.randn(1 << 20, device="cuda", generator=g)
        b = torch.randn(12345, device="cuda", generator=g).bfloat16()
        a0, b0 = a.clone(), b.clone()
        c.allreduce_buckets([a, b])
        c.broadcast(a, 0)
        ga = c.allgather(b)
        rows = torch.arange(7 * 5, device="cuda").view(7, 5)                  # int64 rows: the masks of the negatives
        gv = c.allgather_var(rows, [7])
        ge = c.allgather_var(rows[:0], [0])
        torch.cuda.synchronize()
        assert torch.equal(a, a0) and torch.equal(b, b0) and ga.shape == (1, 12345) and torch.equal(ga[0], b0)
        assert torch.equal(gv, rows) and ge.shape == (0, 5)
        # captured: x -> 2x on the main stream, all-reduce of x on the side stream (fork / join), then x + 1
        x = torch.ones(1 << 16, device="cuda")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, stream=s):
                x.mul_(2.0)
                c.fork()
                c.allreduce_buckets([x], stream=c.stream)
                c.join()
                x.add_(1.0)
        torch.cuda.current_stream().wait_stream(s)
        x.fill_(1.0)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(x, torch.full_like(x, 15.0))          # ((1*2+1)*2+1)*2+1
        # error behaviour of the boundary: bad arguments come back as codes with a message, nothing is enqueued
        from hero_amd import _lib as L
        import ctypes as C
        arr = (L.CommBucket * 1)()
        assert L.lib().hero_comm_allreduce_buckets(c._h, arr, 1, None) != 0 and b"bad bucket" in L.lib().hero_last_error()
        assert L.lib().hero_comm_broadcast(c._h, L.ptr(a), 4, 3, None) != 0
        assert L.lib().hero_comm_allreduce_buckets(None, arr, 1, None) != 0
    finally:
        c.close()


@pytest.mark.parametrize("wire", ["none", "bf16"])
def test_one_rank_gradient_exchange_through_the_c_abi(tmp_path, wire):
    """`set_exchange("abi")`: the gradient buckets of the data-parallel step travel through hero_comm_allreduce_buckets on the
    communicator's side stream (forked at the bucket's finality point, joined in finish()) instead of the process group;
    the worker checks bucketed all-reduce == sum of the local gradients and that optimiser steps stay finite."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HERO_DP_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29561",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), wire, "nccl", "abi"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(tmp_path / "rank0.json"))
    assert res["exchange"] == "abi" and res["collectives"] and res["buckets"] > 3


def test_bench_one_rank_exchange_through_the_c_abi_captured_in_a_hipgraph():
    """`bench.py --exchange abi`: eager run, then the step captured WITH its hero_comm all-reduces; the comm block names
    the exchange."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HERO_DP_FORCE_COLLECTIVES="1",
               HERO_DP_GRAPH="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29563", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--exchange", "abi"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("hero_comm one rank:", out["config"]["launch"], out["ms_per_step"], out["comm"])
    assert out["comm"]["exchange"].startswith("hero_comm") and out["comm"]["ranks_seen"] == 1
    assert out["final_loss"] == out["final_loss"] and out["value"] > 0
    assert out["comm"]["graph_ms_per_step"] is not None and out["config"]["launch"].startswith(("hipGraph replay (step captured", "eager"))
