"""bench.py's launcher on a machine without a GPU: `--gpus N` with no launcher
environment starts N rank processes itself (before anything touches the GPU);
under torch.distributed.run it is a rank.  --rehearse-launch replaces the GPU
work by a gloo all-reduce of the ranks' numbers."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_self_launch_starts_one_process_per_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = _last_json(out.stdout)
    _check_rank_reports(r, 2)
    assert r["backend"] == "gloo"                            # the default: no RCCL needed for barriers
    assert r["shards"] == [[0, 16384], [16384, 32768]]       # contiguous, equal shards (weak scaling)


def _check_rank_reports(r, n):
    """The keys a multi-GPU run's JSON carries so that a reader can see N distinct ranks took part."""
    assert r["ranks_seen"] == n and len(r["ranks"]) == n
    assert [x["rank"] for x in r["ranks"]] == list(range(n))
    assert [x["local_rank"] for x in r["ranks"]] == list(range(n))
    assert len({x["pid"] for x in r["ranks"]}) == n          # one process per rank
    for x in r["ranks"]:
        for key in ("host", "device_index", "device_name", "device_id", "shard", "kernel_ms"):
            assert key in x


def test_under_torch_distributed_run_it_is_a_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--rehearse-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    _check_rank_reports(_last_json(out.stdout), 2)


def test_backend_falls_back_when_the_first_choice_fails():
    # without a GPU, nccl cannot initialise: the group comes up over gloo and says so
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch",
                          "--backend", "nccl"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = _last_json(out.stdout)
    assert r["backend"] == "gloo" and r["ranks_seen"] == 2
    assert r["backend_notes"] and r["backend_notes"][0].startswith("nccl:")     # why the first choice was dropped


def test_launch_wall_limit_ends_the_ranks():
    # ranks that never finish (they sleep): the launcher ends them at its wall limit and exits non-zero
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-launch",
                          "--rehearse-sleep", "120", "--launch-timeout", "8"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 124, (out.returncode, out.stderr)
    assert time.monotonic() - t0 < 60
    assert "--launch-timeout" in out.stderr


def test_a_failing_rank_ends_the_launch():
    # without a GPU every rank of a real run exits with an error; the launcher must return it, not hang
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr
