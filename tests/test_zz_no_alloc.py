"""bd_analyze_batch_noalloc (the entry point for callers that cannot allocate device
memory) against bd_analyze_batch, through the C ABI: identical beat lists, on calls
made one after the other that share the library's fixed device array -- the second
with a smaller input, the third with another strategy -- and a clear error beyond
the fixed capacity."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 44100


def _raw_call(bd, torch, kind, samples, n, L, max_beats, stream=None):
    """The entry point through ctypes on tensors allocated beforehand (poisoned, so that stale
    contents cannot pass for results)."""
    n_streams, stride = samples.shape
    dev = samples.device
    count = torch.full((n_streams,), -7, dtype=torch.int32, device=dev)
    idx = torch.full((n_streams, max_beats), -7, dtype=torch.int32, device=dev)
    ms = torch.full((n_streams, max_beats), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    st = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
    rc = bd.lib().bd_analyze_batch_noalloc(kind.value, SR, None, samples.data_ptr(), n_streams, n, stride, L,
                                           count.data_ptr(), idx.data_ptr(), ms.data_ptr(), max_beats, st)
    assert rc == 0, bd.lib().bd_last_error().decode()
    torch.cuda.synchronize()
    return count.cpu().numpy(), idx.cpu().numpy(), ms.cpu().numpy()


def _expect(bd, torch, kind, samples, n, L, max_beats):
    res = bd.analyze_batch(kind, SR, samples, L, n_samples=n, max_beats=max_beats)
    torch.cuda.synchronize()
    return res.beat_count.cpu().numpy(), res.beat_idx.cpu().numpy(), res.beat_ms.cpu().numpy()


def _same(got, want, max_beats):
    gc, gi, gm = got
    wc, wi, wm = want
    assert (gc == wc).all()
    assert wc.sum() > 0
    for s in range(len(wc)):
        k = min(int(wc[s]), max_beats)
        assert (gi[s, :k] == wi[s, :k]).all() and (gm[s, :k] == wm[s, :k]).all(), s


def test_noalloc_matches_analyze_batch_on_consecutive_calls(bd):
    import torch
    dev = torch.device("cuda:0")
    # call 1: 96 streams x 20 s, Spectrum, 512-sample callbacks
    n1 = SR * 20
    big = torch.zeros((96, (n1 + 63) & ~63), dtype=torch.int16, device=dev)
    bd.synth_fill(big, n_samples=n1, seed=11)
    # call 2 (smaller input, shorter rows: the decisions of call 1 are still in the array beyond them)
    n2 = SR * 7 + 123
    small = torch.zeros((17, (n2 + 7) & ~7), dtype=torch.int16, device=dev)
    bd.synth_fill(small, n_samples=n2, seed=12, first_stream=1000)
    torch.cuda.synchronize()
    # (96: granules of 32 samples, 900: the window kernel, 1536: granules of 64, 7: the generic kernel)
    for kind, L in ((bd.StrategyKind.Spectrum, 512), (bd.StrategyKind.LPF, 1024), (bd.StrategyKind.Spectrum, 441),
                    (bd.StrategyKind.Spectrum, 529), (bd.StrategyKind.Spectrum, 96), (bd.StrategyKind.Spectrum, 900),
                    (bd.StrategyKind.Spectrum, 1536), (bd.StrategyKind.LPF, 441)):
        mb1 = bd.batch_max_beats(kind, SR, n1, L)
        mb2 = bd.batch_max_beats(kind, SR, n2, L)
        want1 = _expect(bd, torch, kind, big, n1, L, mb1)
        want2 = _expect(bd, torch, kind, small, n2, L, mb2)
        got1 = _raw_call(bd, torch, kind, big, n1, L, mb1)
        got2 = _raw_call(bd, torch, kind, small, n2, L, mb2)      # second call in a row, smaller input
        _same(got1, want1, mb1)
        _same(got2, want2, mb2)
        # and once more the large one after the small one
        _same(_raw_call(bd, torch, kind, big, n1, L, mb1), want1, mb1)


def test_noalloc_just_under_capacity_uses_dense_rows(bd):
    """A batch whose decision rows fit the fixed array only when packed densely (rows padded to 128-byte lines
    would not): 18,000 streams x 5,512 callbacks of 16 samples = 99.2 M of the 100.7 M callbacks the array holds.
    The unaligned dense rows go through a strategy kernel and the refractory pass; streams from both ends and the
    middle of the batch must equal bd_analyze_batch on those streams alone."""
    import torch
    dev = torch.device("cuda:0")
    n_streams, L = 18000, 16
    n = SR * 2                             # 5,512 callbacks: two seconds, a few clicks in every stream
    n_cb = n // L
    cap = bd.NOALLOC_MAX_CALLBACKS
    assert n_streams * ((n_cb + 127) & ~127) > cap >= n_streams * n_cb
    x = torch.zeros((n_streams, (n + 7) & ~7), dtype=torch.int16, device=dev)
    bd.synth_fill(x, n_samples=n, seed=14)
    torch.cuda.synchronize()
    for kind in (bd.StrategyKind.Spectrum, bd.StrategyKind.LPF):
        mb = bd.batch_max_beats(kind, SR, n, L)
        gc, gi, gm = _raw_call(bd, torch, kind, x, n, L, mb)
        # (the low-pass strategy finds no beat in windows of 16 samples, here as in the reference: its rows of
        # zeros are compared all the same)
        assert gc.sum() > 100000 or kind == bd.StrategyKind.LPF
        for lo in (0, n_streams // 2 - 100, n_streams - 200):
            wc, wi, wm = _expect(bd, torch, kind, x[lo:lo + 200], n, L, mb)
            assert (gc[lo:lo + 200] == wc).all(), (kind, lo)
            for s in range(200):
                k = min(int(wc[s]), mb)
                assert (gi[lo + s, :k] == wi[s, :k]).all() and (gm[lo + s, :k] == wm[s, :k]).all(), (kind, lo, s)
        # a small call right after it (the array is full of the large call's decisions)
        small = x[:9]
        if gc[:9].sum():
            _same(_raw_call(bd, torch, kind, small, n, L, mb), _expect(bd, torch, kind, small, n, L, mb), mb)
    del x
    torch.cuda.empty_cache()


def test_noalloc_on_a_side_stream_and_truncated_lists(bd):
    import torch
    dev = torch.device("cuda:0")
    n = SR * 12
    x = torch.zeros((33, (n + 63) & ~63), dtype=torch.int16, device=dev)
    bd.synth_fill(x, n_samples=n, seed=13)
    torch.cuda.synchronize()
    kind, L = bd.StrategyKind.Spectrum, 1024
    want = _expect(bd, torch, kind, x, n, L, 3)             # lists cut at 3 beats, counts complete
    side = torch.cuda.Stream(device=dev)
    got = _raw_call(bd, torch, kind, x, n, L, 3, stream=side)
    _same(got, want, 3)
    assert int(want[0].max()) > 3


def test_noalloc_rejects_what_does_not_fit_and_bad_arguments(bd):
    import torch
    L_ = bd.lib()
    dev = torch.device("cuda:0")
    x = torch.zeros((4, 4096), dtype=torch.int16, device=dev)
    count = torch.zeros((4,), dtype=torch.int32, device=dev)
    cap = bd.NOALLOC_MAX_CALLBACKS
    # 4 streams whose callbacks together exceed the capacity: rejected before anything is launched or read
    n_big = (cap // 4 + 1) * 64
    rc = L_.bd_analyze_batch_noalloc(1, SR, None, x.data_ptr(), 4, n_big, n_big, 64, count.data_ptr(), None, None, 0, None)
    assert rc == bd.api.BD_ERR_UNSUPPORTED
    assert b"capacity" in L_.bd_last_error()
    assert L_.bd_analyze_batch_noalloc(1, SR, None, x.data_ptr(), 4, 4096, 4096, 0, count.data_ptr(), None, None, 0, None) \
        == bd.api.BD_ERR_EMPTY_WINDOW
    assert L_.bd_analyze_batch_noalloc(1, SR, None, x.data_ptr(), 4, 4096, 4096, 512, None, None, None, 0, None) \
        == bd.api.BD_ERR_INVALID_ARG
    assert L_.bd_analyze_batch_noalloc(1, SR, None, x.data_ptr(), 4, 4096, 1024, 512, count.data_ptr(), None, None, 0, None) \
        == bd.api.BD_ERR_INVALID_ARG
    # counts only (no lists), and a batch without any complete callback
    assert L_.bd_analyze_batch_noalloc(1, SR, None, x.data_ptr(), 4, 4096, 4096, 512, count.data_ptr(), None, None, 0, None) == 0
    assert L_.bd_analyze_batch_noalloc(0, SR, None, x.data_ptr(), 4, 100, 4096, 512, count.data_ptr(), None, None, 0, None) == 0
    torch.cuda.synchronize()
    assert (count.cpu().numpy() == 0).all()


def test_noalloc_inside_a_captured_graph(bd):
    """What the entry point is for: captured into a HIP graph (no allocation may happen there) and
    replayed on new samples in the same buffers."""
    import torch
    dev = torch.device("cuda:0")
    n, L = SR * 9, 529
    kind = bd.StrategyKind.Spectrum
    mb = bd.batch_max_beats(kind, SR, n, L)
    x = torch.zeros((21, (n + 63) & ~63), dtype=torch.int16, device=dev)
    count = torch.zeros((21,), dtype=torch.int32, device=dev)
    idx = torch.zeros((21, mb), dtype=torch.int32, device=dev)
    ms = torch.zeros((21, mb), dtype=torch.int32, device=dev)
    bd.synth_fill(x, n_samples=n, seed=21)
    _raw_call(bd, torch, kind, x, n, L, mb)             # first call on the device happens outside the capture
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = bd.lib().bd_analyze_batch_noalloc(kind.value, SR, None, x.data_ptr(), 21, n, x.shape[1], L, count.data_ptr(),
                                               idx.data_ptr(), ms.data_ptr(), mb, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, bd.lib().bd_last_error().decode()
    for seed in (21, 22, 23):
        bd.synth_fill(x, n_samples=n, seed=seed)
        count.fill_(-7); idx.fill_(-7); ms.fill_(-7)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = _expect(bd, torch, kind, x, n, L, mb)
        _same((count.cpu().numpy(), idx.cpu().numpy(), ms.cpu().numpy()), want, mb)
