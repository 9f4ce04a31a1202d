"""Chunked executor (bd_chunked_*): a shard of streams analysed a chunk at a time,
the next chunk produced on a side stream.  Streams are independent in the
reference (src/strategies/mod.rs:39-56), so the beat / onset lists must equal
those of one launch over the whole shard, whatever the chunk size; at the size of
one GPU's share of the 131,072-track job (16,384 five-minute tracks, 433 GB of
samples that never exist at once) tracks sampled from every chunk are checked
against the CPU definition of the analysis."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SR = 44100


def _lists(count, idx, ms):
    count, idx, ms = count.cpu().numpy(), idx.cpu().numpy(), ms.cpu().numpy()
    return [list(zip(idx[s, :min(count[s], idx.shape[1])].tolist(), ms[s, :min(count[s], idx.shape[1])].tolist()))
            for s in range(len(count))], count


def _outputs(torch, n_streams, max_events):
    dev = torch.device("cuda:0")
    return (torch.full((n_streams,), -1, dtype=torch.int32, device=dev),
            torch.full((n_streams, max_events), -1, dtype=torch.int32, device=dev),
            torch.full((n_streams, max_events), -1, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("chunk", [8, 16, 37, 64])
def test_chunked_strategies_equal_one_launch(bd, chunk):
    import torch
    n_streams, n, seed, first = 37, SR * 5 + 321, 21, 500
    whole = torch.zeros((n_streams, (n + 63) & ~63), dtype=torch.int16, device="cuda:0")
    bd.synth_fill(whole, n_samples=n, seed=seed, first_stream=first)
    ex = bd.ChunkedExecutor(chunk, n)
    try:
        # several passes through the same executor, in a row: buffers and events are reused
        for kind, L in ((bd.StrategyKind.Spectrum, 512), (bd.StrategyKind.LPF, 1024), (bd.StrategyKind.Spectrum, 441)):
            mb = bd.batch_max_beats(kind, SR, n, L)
            want = bd.analyze_batch(kind, SR, whole, L, n_samples=n, max_beats=mb)
            count, idx, ms = _outputs(torch, n_streams, mb)
            ex.analyze(kind, SR, L, n_streams, count, idx, ms, first_stream=first, seed=seed)
            torch.cuda.synchronize()
            got, gc = _lists(count, idx, ms)
            assert (gc == want.beat_count.cpu().numpy()).all()
            assert gc.sum() > 0
            assert got == [want.beats(s) for s in range(n_streams)]
        assert ex.device_bytes() >= 2 * chunk * n * 2
    finally:
        ex.close()


def test_chunked_passes_on_alternating_streams(bd):
    """Consecutive passes of one executor on two different HIP streams, nothing synchronised in between: the
    intermediate buffer they share must not be rewritten by one pass while the other still uses it."""
    import torch
    n_streams, n, seed = 48, SR * 8 + 77, 31
    whole = torch.zeros((n_streams, (n + 63) & ~63), dtype=torch.int16, device="cuda:0")
    bd.synth_fill(whole, n_samples=n, seed=seed)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ex = bd.ChunkedExecutor(5, n)
    try:
        jobs = []
        for i, (kind, L) in enumerate(((bd.StrategyKind.Spectrum, 512), (bd.StrategyKind.LPF, 1024),
                                       (bd.StrategyKind.Spectrum, 529), (bd.StrategyKind.LPF, 441)) * 2):
            mb = bd.batch_max_beats(kind, SR, n, L)
            out = _outputs(torch, n_streams, mb)
            ex.analyze(kind, SR, L, n_streams, *out, seed=seed, stream=streams[i % 2])
            jobs.append((kind, L, mb, out))
        torch.cuda.synchronize()
        for kind, L, mb, (count, idx, ms) in jobs:
            want = bd.analyze_batch(kind, SR, whole, L, n_samples=n, max_beats=mb)
            got, gc = _lists(count, idx, ms)
            assert got == [want.beats(s) for s in range(n_streams)], (kind, L)
    finally:
        ex.close()


def test_chunked_onset_equals_one_launch_and_caller_sources(bd):
    import torch
    n_streams, n, seed = 21, SR * 6, 22
    whole = torch.zeros((n_streams, (n + 63) & ~63), dtype=torch.int16, device="cuda:0")
    bd.synth_fill(whole, n_samples=n, seed=seed)
    p = bd.OnsetParams.default(fft_len=2048)
    want = bd.onset_batch(SR, whole, n_samples=n, params=p, max_onsets=256)
    torch.cuda.synchronize()
    ex = bd.ChunkedExecutor(5, n)
    try:
        count, frame, ms = _outputs(torch, n_streams, 256)
        ex.onset(SR, n_streams, count, frame, ms, params=p, seed=seed)
        torch.cuda.synchronize()
        got, gc = _lists(count, frame, ms)
        assert gc.sum() > 0
        assert got == [want.onsets(s) for s in range(n_streams)]

        # a caller's source (a Python callable behind the C callback): enqueues the generator itself
        calls = []

        def source(dst, first, rows, n_samples, stride, st):
            calls.append((first, rows))
            return bd.lib().bd_synth_fill(dst, rows, n_samples, stride, first, seed, SR, st)

        count, frame, ms = _outputs(torch, n_streams, 256)
        ex.onset(SR, n_streams, count, frame, ms, params=p, source=source)
        torch.cuda.synchronize()
        assert calls == [(0, 5), (5, 5), (10, 5), (15, 5), (20, 1)]
        assert _lists(count, frame, ms)[0] == got

        # a failing source ends the pass with its status
        with pytest.raises(bd.BeatDetectorError) as e:
            ex.onset(SR, n_streams, count, frame, ms, params=p, source=lambda *a: bd.api.BD_ERR_UNSUPPORTED)
        assert e.value.status == bd.api.BD_ERR_UNSUPPORTED
        torch.cuda.synchronize()
        # ... and the executor still works afterwards
        count2, idx2, ms2 = _outputs(torch, n_streams, 64)
        ex.analyze(bd.StrategyKind.LPF, SR, 1024, n_streams, count2, idx2, ms2, seed=seed)
        torch.cuda.synchronize()
        w = bd.analyze_batch(bd.StrategyKind.LPF, SR, whole, 1024, n_samples=n, max_beats=64)
        assert _lists(count2, idx2, ms2)[0] == [w.beats(s) for s in range(n_streams)]
    finally:
        ex.close()


def test_chunked_argument_errors(bd):
    L_ = bd.lib()
    h = C.c_void_p()
    assert L_.bd_chunked_create(0, 1000, -1, C.byref(h)) == bd.api.BD_ERR_INVALID_ARG
    assert L_.bd_chunked_create(4, 0, -1, C.byref(h)) == bd.api.BD_ERR_INVALID_ARG
    assert L_.bd_chunked_create(4, 1000, 99, C.byref(h)) == bd.api.BD_ERR_INVALID_ARG
    assert L_.bd_chunked_create(4, 1000, -1, None) == bd.api.BD_ERR_INVALID_ARG
    assert L_.bd_chunked_device_bytes(None) == 0
    L_.bd_chunked_destroy(None)
    ex = bd.ChunkedExecutor(4, 4096)
    import torch
    count, idx, ms = _outputs(torch, 4, 8)
    with pytest.raises(bd.BeatDetectorError):
        ex.analyze(bd.StrategyKind.LPF, SR, 0, 4, count, idx, ms)
    with pytest.raises(bd.BeatDetectorError):
        ex.analyze(bd.StrategyKind.LPF, 0, 1024, 4, count, idx, ms)
    ex.analyze(bd.StrategyKind.LPF, SR, 1024, 0, count, idx, ms)   # an empty shard is fine
    ex.close()


def test_one_gpu_share_of_the_131072_track_job(bd, oracle_ext):
    """16,384 tracks x 300 s, 2048-point STFT, flux, peak picking: 4 chunks of 4,096 tracks."""
    import torch
    torch.cuda.empty_cache()
    n_streams, chunk, n, seed, first = 16384, 4096, SR * 300, 0x5EED, 3 * 16384   # the share of rank 3 of 8
    p = bd.OnsetParams.default(fft_len=2048)
    max_onsets = 3001
    ex = bd.ChunkedExecutor(chunk, n)
    try:
        assert ex.device_bytes() > 2 * chunk * n * 2
        count, frame, ms = _outputs(torch, n_streams, max_onsets)
        ex.onset(SR, n_streams, count, frame, ms, params=p, first_stream=first, seed=seed)
        torch.cuda.synchronize()
        got, gc = _lists(count, frame, ms)
        assert (gc > 0).all() and (gc <= max_onsets).all()
        po = oracle_ext.default_params(fft_len=2048)
        for c in range(n_streams // chunk):
            for s in (c * chunk, c * chunk + 1777, (c + 1) * chunk - 1):
                row = bd.synth_fill_host(1, n, first_stream=first + s, seed=seed)[0]
                flux, _ = oracle_ext.onset_stft(row, SR, po)
                _, f, t = oracle_ext.onset_pick(flux, SR, po)
                assert got[s] == list(zip(f.tolist(), t.tolist())), s
    finally:
        ex.close()
