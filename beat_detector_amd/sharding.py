"""Stream sharding for multi-GPU runs.  Streams are independent (no state
crosses streams: the reference keeps AnalysisState and the ring per detector,
src/strategies/mod.rs:39-56, spectrum.rs:38-44), so each rank owns a contiguous
range of streams and the data path needs no collective; only tiny per-stream
results (beat counts, beat lists) are gathered."""
from __future__ import annotations

from typing import List, Tuple


def shard_range(n_streams: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, last) streams of `rank`: the same contiguous split as
    bd_analyze_batch_multi (first = n * rank / world)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return n_streams * rank // world, n_streams * (rank + 1) // world


def shard_ranges(n_streams: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n_streams, r, world) for r in range(world)]


def gather_counts(local_counts, dist, world: int, n_streams: int):
    """All-gathers per-stream beat counts of contiguous shards into one tensor
    of n_streams entries (same on every rank).  `local_counts` is a 1-D int32
    torch tensor on the backend's device."""
    import torch
    sizes = [b - a for a, b in shard_ranges(n_streams, world)]
    pad = max(sizes) if sizes else 0
    buf = torch.zeros(pad, dtype=local_counts.dtype, device=local_counts.device)
    buf[: local_counts.numel()] = local_counts
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:n] for o, n in zip(out, sizes)])
