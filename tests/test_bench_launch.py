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
    assert r["ranks"] == 2 and r["rank_sum"] == 1 and r["local_rank_sum"] == 1
    assert r["shards"] == [[0, 16384], [16384, 32768]]       # contiguous, equal shards (weak scaling)


def test_under_torch_distributed_run_it_is_a_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--rehearse-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert _last_json(out.stdout)["ranks"] == 2


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
