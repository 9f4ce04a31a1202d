"""Multi-rank plumbing on CPU (gloo, world size 2): contiguous stream shards, no
collective on the data path, tiny results gathered.  The per-rank analysis is
done by the oracle here (no GPU in this container); on the GPU box the same
sharding drives bench.py, one process per GPU over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from beat_detector_amd import sharding


def test_shard_ranges_cover_all_streams_once():
    for n in (0, 1, 7, 8, 16384, 131072 + 5):
        for w in (1, 2, 3, 4, 8):
            r = sharding.shard_ranges(n, w)
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, n, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import beat_detector_amd as bd
    from oracle import bd_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = sharding.shard_range(n_streams, rank, world)
    # every rank synthesises and analyses only its own streams
    x = bd.synth_fill_host(b - a, n, first_stream=a, seed=5)
    _, counts = bd_oracle.run_batch(bd_oracle.KIND_SPECTRUM, 44100, x, 512)
    allc = sharding.gather_counts(torch.from_numpy(counts.astype(np.int32)), dist, world, n_streams)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the timing reduction bench.py uses
    if rank == 0:
        np.save(os.path.join(out_dir, "counts.npy"), allc.numpy())
        np.save(os.path.join(out_dir, "tmax.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    import beat_detector_amd as bd
    from oracle import bd_oracle
    n_streams, n, world = 5, 44100 * 2, 2
    mp.spawn(_worker, args=(world, _free_port(), n_streams, n, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "counts.npy")
    x = bd.synth_fill_host(n_streams, n, first_stream=0, seed=5)
    _, want = bd_oracle.run_batch(bd_oracle.KIND_SPECTRUM, 44100, x, 512)
    np.testing.assert_array_equal(got, want.astype(np.int32))
    assert np.load(tmp_path / "tmax.npy")[0] == 2.0
