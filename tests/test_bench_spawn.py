"""bench.py starts its own ranks: `python bench.py --gpus N` with no launcher around it spawns N
child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), the ranks meet
over the TCP rendezvous, shard the calculation, exchange slab rows and rank 0's single JSON line
is relayed.  Driven here with the orchestration-only stub engine (no GPU in this container); the
GPU path differs only in the engine object."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(*extra, env=None):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--stub-engine", "--config", "C1", "--steps", "2", "--warmup", "1", *extra]
    full_env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        full_env.pop(k, None)
    full_env.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=full_env)


@pytest.mark.parametrize("shard, n", [("k", 2), ("frames", 2), ("auto", 3)])
def test_bench_spawns_its_ranks_and_prints_one_line(shard, n):
    res = _run("--gpus", str(n), "--shard", shard)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 2 and out["warmup"] == 1
    assert out["data"].startswith("STUB ENGINE") and out["value"] == 0.0     # nothing was measured
    cfg = out["config"]
    assert cfg["transport"] == "host"                                         # the stub has no RCCL
    assert sum(cfg["k_points_per_rank"]) == 32 and len(cfg["k_points_per_rank"]) == n
    assert cfg["shard_mode"] == ("k" if shard == "k" else "frames")           # 32 k over n ranks: auto -> frames
    assert out["scaling"] == "strong" and out["higher_is_better"] is True
    # what makes the N > 1 line explain itself (round 3): every rank's own stage times beside the
    # per-stage maximum, why RCCL was not used, the drop-in call on the root, the intensity-only
    # gather, and the north star's k-partitioning measured beside frame sharding when auto picked that
    stage_names = {"h2d", "phase", "project", "fft", "epilogue", "gather", "transpose", "d2h"}
    assert set(out["stages_ms_per_step"]) == stage_names
    assert len(out["stages_ms_per_step_by_rank"]) == n
    assert all(set(r) == stage_names | {"k1_avg_launch_ms"} for r in out["stages_ms_per_step_by_rank"])
    assert cfg["transport_error"] == "stub engine: no RCCL" and "NOT xGMI" in cfg["transport_note"]
    e2e = out["end_to_end"]
    assert e2e["ms"] >= e2e["calculate_only_ms"] > 0 and e2e["result_bytes"] == 28 * 4096 * 32
    inten = out["variants"]["intensity_only"]
    assert inten["shard_mode"] == cfg["shard_mode"] and inten["ms_per_step"] > 0
    if shard == "k":
        assert "shard_k" not in out["variants"]
    else:
        sk = out["variants"]["shard_k"]
        assert sk["shard_mode"] == "k" and sum(sk["k_points_per_rank"]) == 32 and len(sk["k_points_per_rank"]) == n


def test_bench_runs_as_one_rank_under_a_launcher():
    """With WORLD_SIZE exported (torch.distributed.run, or the spawner itself) bench.py IS a rank."""
    res = _run("--gpus", "1", env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip())
    assert out["n_gpus"] == 1 and out["config"]["parallelism"] == "single GPU"
    assert "stages_ms_per_step_by_rank" not in out and "transport_error" not in out["config"]


def test_a_failing_rank_fails_the_launch():
    res = _run("--gpus", "2", "--config", "C9")            # argparse rejects it in every rank
    assert res.returncode != 0 and not res.stdout.strip()


def test_bench_under_torch_distributed_run():
    """The launch line the driver uses for N > 1: torch.distributed.run exports RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*, bench.py runs as one of its ranks (it does not import torch itself)."""
    pytest.importorskip("torch")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--stub-engine", "--config", "C1",
           "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["k_points_per_rank"] == [16, 16]
