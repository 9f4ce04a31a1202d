"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same
inputs.  Integer results (decisions, beat indices, beat times, LPF values,
converters, generator) are compared bit-exactly; Spectrum band values within
1e-4 relative (see util.value_tolerance)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

SR = 44100


def _torch():
    import torch
    return torch


def _run_gpu(bd, kind, sr, x, L, params=None):
    torch = _torch()
    n_streams, n = x.shape
    stride = (n + 7) & ~7
    buf = torch.zeros((n_streams, stride), dtype=torch.int16, device="cuda")
    buf[:, :n] = torch.from_numpy(x).cuda()
    res = bd.analyze_batch(kind, sr, buf, L, n_samples=n, params=params)
    # decisions only (no value rows): the same decisions and beat lists from every kernel
    bare = bd.analyze_batch(kind, sr, buf, L, n_samples=n, params=params, want_values=False)
    torch.cuda.synchronize()
    assert bare.values is None
    assert torch.equal(bare.flags, res.flags) and torch.equal(bare.beat_count, res.beat_count)
    assert torch.equal(bare.beat_idx, res.beat_idx) and torch.equal(bare.beat_ms, res.beat_ms)
    return res


def _compare(bd, oracle, kind_name, sr, x, L, bd_params=None, o_params=None, exact_values=False):
    kind = bd.StrategyKind[kind_name]
    okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
    res = _run_gpu(bd, kind, sr, x, L, bd_params)
    ref_v, ref_c, ref_b = util.oracle_batch(oracle, okind, sr, x, L, o_params)
    got_v = res.values.cpu().numpy()
    got_c = res.flags.cpu().numpy()
    assert got_v.shape == ref_v.shape
    if exact_values:
        np.testing.assert_array_equal(got_v, ref_v)
        adjacent = np.zeros_like(ref_c, dtype=bool)
    else:
        p = o_params if o_params is not None else oracle.default_params()
        nb, _, _ = oracle.bin_range(sr, p)
        div = {0: nb, 1: 1024, 2: 0}[p.spectrum_divisor]
        bad = ~util.values_close(got_v, ref_v, np.abs(x.astype(np.int32)).max(), div)
        assert not bad.any(), f"{bad.sum()} band values off; worst {np.abs(got_v - ref_v).max()}"
        adjacent = util.threshold_adjacent(ref_v, p.spectrum_threshold)
    mism = (got_c != ref_c)
    assert not (mism & ~adjacent).any(), f"decisions differ at {np.argwhere(mism & ~adjacent)[:10]}"
    if not mism.any():
        # beat lists must then be identical: index and relative_ms
        for s in range(x.shape[0]):
            assert res.beats(s) == ref_b[s], f"stream {s}"
    else:
        print("threshold-adjacent callbacks with flipped decision:", np.argwhere(mism).tolist())
    return res, ref_v, ref_c, ref_b


def _synth(bd, n_streams, n, first=0, seed=0x5EED, sr=SR):
    return bd.synth_fill_host(n_streams, n, first_stream=first, seed=seed, sampling_rate=sr)


@pytest.mark.parametrize("L", [512, 1024])
def test_spectrum_fast_path(bd, oracle, L):
    x = _synth(bd, 6, SR * 12)
    res, ref_v, ref_c, ref_b = _compare(bd, oracle, "Spectrum", SR, x, L)
    assert bd.last_kernel_name() == "bd_spectrum_band_kernel"
    assert ref_c.sum() > 0 and sum(len(b) for b in ref_b) > 0   # the case exercises beats
    assert (ref_c == 0).sum() > 0


@pytest.mark.parametrize("L", [1536, 2048, 2560, 3072, 3584, 4096, 4608, 5120, 5632, 6144, 6656, 7168, 7680, 8192])
def test_spectrum_fast_path_long_callbacks(bd, oracle, knobs, L):
    """Callbacks of 3 .. 16 half-blocks: the ring is the callback's last two half-blocks, the gate looks at all
    of them.  Streams of one callback, of a few, of several runs with a partial last one; rows whose length is
    no multiple of the callback; the unpinned choices; a batch that shortens the runs; other run lengths."""
    for n, streams in ((L, 2), (3 * L + 5, 3), (L * 67 + 1001, 2), (L * 3 - 1, 1), (512 * 131, 2)):
        x = _synth(bd, streams, n, first=7)
        _compare(bd, oracle, "Spectrum", SR, x, L)
        if n >= L:
            assert bd.last_kernel_name() == "bd_spectrum_band_kernel"
    x = _synth(bd, 3, L * 40 + 9, first=21)
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2),
               dict(spectrum_max_hz=50.0), dict(spectrum_max_hz=10.0)):
        _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**kw), oracle.default_params(**kw))
    m = L // 512
    unit = next(u for u in range(m, 200, m) if u % 4 == 0)
    for run in (unit, 2 * unit):
        if run <= 128:
            knobs.set("BD_SPEC_RUN", str(run))
            _compare(bd, oracle, "Spectrum", SR, x, L)
    knobs.unset("BD_SPEC_RUN")
    # a loud sample in each of the callback's half-blocks in turn opens the gate (quiet stream otherwise)
    quiet = (_synth(bd, 1, L * 4, first=3) // 64).astype(np.int16)
    kw = dict(spectrum_threshold=10.0, min_beat_gap_ms=0)    # every callback whose gate is open is a beat
    for h in range(m):
        y = quiet.copy()
        y[0, 2 * L + 512 * h + 100] = 20000
        res, _, ref_c, _ = _compare(bd, oracle, "Spectrum", SR, y, L, bd.Params.default(**kw), oracle.default_params(**kw))
        assert ref_c[0].tolist() == [0, 0, 1, 0]


@pytest.mark.parametrize("L", [256, 128])
def test_spectrum_fast_path_256_and_128_sample_callbacks(bd, oracle, knobs, L):
    """256- and 128-sample callbacks through the half-block kernel: two / four callbacks per half-block, the ring
    is four quarter-blocks / eight eighth-blocks.  Streams of one callback, of an odd number (the last half-block is half a callback's), of
    several runs with a partial last one, rows whose length is no multiple of the callback; the unpinned choices;
    other run lengths; other sampling rates (one and no non-DC bin); the gate looks at the callback's own samples
    only; the granule kernel on the same data."""
    torch = _torch()
    for n, streams in ((L, 2), (3 * L + 5, 3), (512 * 67 + 300, 2), (L * 2 - 1, 1), (512 * 131, 2), (512 * 64 + 256, 3),
                       (512 * 128 + 255, 1), (512 * 64 + 128, 1), (512 * 65 + 384 + 17, 2), (5 * L, 1), (7 * L + 3, 1)):
        x = _synth(bd, streams, n, first=7)
        _compare(bd, oracle, "Spectrum", SR, x, L)
        if n >= L:
            assert bd.last_kernel_name() == "bd_spectrum_band_kernel"
    x = _synth(bd, 3, L * 300 + 9, first=21)
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2),
               dict(spectrum_max_hz=50.0), dict(spectrum_max_hz=10.0)):
        _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**kw), oracle.default_params(**kw))
    for run in ("8", "16", "32", "64", "128"):          # 128 is more than the slots hold: the launcher keeps its own
        knobs.set("BD_SPEC_RUN", run)
        _compare(bd, oracle, "Spectrum", SR, x, L)
    knobs.unset("BD_SPEC_RUN")
    for sr in (48000, 96000):
        y = _synth(bd, 2, 60000, first=5, sr=sr)
        _compare(bd, oracle, "Spectrum", sr, y, L)
        assert bd.last_kernel_name() == "bd_spectrum_band_kernel"
    # a loud sample opens the gate of the callback it lies in and of no other (quiet stream otherwise)
    quiet = (_synth(bd, 1, L * 12, first=3) // 64).astype(np.int16)
    kw = dict(spectrum_threshold=10.0, min_beat_gap_ms=0)    # every callback whose gate is open is a beat
    for cb in (4, 5, 6, 7):
        y = quiet.copy()
        y[0, cb * L + 100] = 20000
        _, _, ref_c, _ = _compare(bd, oracle, "Spectrum", SR, y, L, bd.Params.default(**kw), oracle.default_params(**kw))
        assert ref_c[0].tolist() == [1 if j == cb else 0 for j in range(12)]
    # both kernels on the same data
    dev = torch.from_numpy(np.ascontiguousarray(np.pad(x, ((0, 0), (0, 64 - x.shape[1] % 64))))).cuda()
    v1, f1 = bd.batch_candidates(bd.StrategyKind.Spectrum, SR, dev, L, n_samples=x.shape[1])
    knobs.set("BD_SPEC_NO_QUARTER", "1")
    v2, f2 = bd.batch_candidates(bd.StrategyKind.Spectrum, SR, dev, L, n_samples=x.shape[1])
    assert bd.last_kernel_name() == "bd_spectrum_granule_kernel"
    knobs.unset("BD_SPEC_NO_QUARTER")
    assert torch.equal(f1, f2)
    assert torch.isclose(v1, v2, rtol=1e-5, atol=1.0).all()


def test_spectrum_ragged_lengths(bd, oracle):
    # lengths that are not multiples of the callback, of a half-block run, or of 8
    for n in (511, 512, 1023, 1024, 1537, 4096 + 3, 512 * 129 + 5, 1024 * 67 + 1001):
        x = _synth(bd, 3, n, first=11)
        for L in (512, 1024):
            _compare(bd, oracle, "Spectrum", SR, x, L)


@pytest.mark.parametrize("L", [1, 7, 50, 61, 769, 790, 800, 882, 1000, 1008, 1023, 1025, 1500, 3000, 4160, 8256, 8704])
def test_spectrum_window_callback_lengths(bd, oracle, L):
    """Callback lengths that are not multiples of 64 (or lie above 4096): the window kernel, with the
    callbacks whose ring starts before the stream on the generic kernel; streams of one callback, of a
    few, of more than one run of 64 callbacks."""
    for n, streams in ((40000, 2), (max(3 * L + 5, 1100), 3), (L, 2), (min(64 * L * 3 + 7, 300000), 1)):
        x = _synth(bd, streams, n, first=3)
        _compare(bd, oracle, "Spectrum", SR, x, L)
        n_cb = n // L
        if (n_cb * L >= 1024):
            assert bd.last_kernel_name() == "bd_spectrum_window_kernel"
        else:
            assert bd.last_kernel_name() == "bd_spectrum_generic_kernel"
    x = _synth(bd, 2, 50000, first=9)
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2),
               dict(spectrum_max_hz=50.0), dict(spectrum_max_hz=10.0)):
        _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**kw), oracle.default_params(**kw))


@pytest.mark.parametrize("L", [65, 66, 70, 80, 87, 95, 97, 100, 101, 112, 119, 125, 127,
                               144, 160, 189, 240, 252, 288, 352, 441, 480, 504, 693, 756,
                               129, 130, 147, 150, 171, 191, 193, 200, 205, 221, 250, 255,
                               257, 270, 280, 281, 300, 317, 319, 333, 341, 342, 350, 383, 385, 399,
                               400, 401, 450, 500, 511, 513, 517, 529, 532, 540, 552, 553, 600, 639,
                               641, 664, 665, 700, 720, 735, 767, 576, 640, 704])
def test_spectrum_split_kernel_callback_lengths(bd, oracle, L):
    """Callback lengths around what the reference's author sees on Linux ("mostly 500-540",
    spectrum.rs:40-42), on no grid: the split kernel (ring = tail of an earlier callback + whole
    callbacks).  Rings of one and of two whole callbacks, heads shorter and longer than a lane's 40
    samples, a cut on a lane boundary (532), tails of two samples (511); streams of one callback, of a
    few, of several runs of 64; rows at odd addresses; the unpinned choices."""
    import torch
    for n, streams in ((40000, 2), (max(3 * L + 5, 1100), 3), (L, 2), (2 * L + 1, 2), (64 * L * 3 + 7, 1), (65 * L, 2),
                       (129 * L + L - 1, 1)):
        x = _synth(bd, streams, n, first=3)
        _compare(bd, oracle, "Spectrum", SR, x, L)
        assert bd.last_kernel_name() == ("bd_spectrum_split_kernel" if n // L > 1 else "bd_spectrum_generic_kernel")
    x = _synth(bd, 2, 50000, first=9)
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2),
               dict(spectrum_max_hz=50.0), dict(spectrum_max_hz=10.0), dict(min_amplitude=0)):
        _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**kw), oracle.default_params(**kw))
    # rows that are only 2-byte aligned, the last stream ending with the buffer
    n = 30011
    x = _synth(bd, 3, n, first=5)
    flat = torch.zeros(1 + 3 * n, dtype=torch.int16, device="cuda")
    view = flat[1:].view(3, n)
    view[:, :] = torch.from_numpy(x).cuda()
    res = bd.analyze_batch(bd.StrategyKind.Spectrum, SR, view, L, n_samples=n)
    ref_v, ref_c, ref_b = util.oracle_batch(oracle, oracle.KIND_SPECTRUM, SR, x, L)
    assert util.values_close(res.values.cpu().numpy(), ref_v, 32768, 3).all()
    np.testing.assert_array_equal(res.flags.cpu().numpy(), ref_c)


@pytest.mark.parametrize("subruns", [2, 3, 8])
def test_spectrum_split_kernel_several_runs_per_wavefront(bd, oracle, knobs, subruns):
    """A wavefront that walks several runs of 64 callbacks carries the sums of the last callbacks from run to
    run instead of reading them again (small batches use one run per wavefront, so the switch forces it):
    every flavour of the kernel, streams that end inside a run, on a run boundary and one callback after it,
    rows at odd addresses."""
    import torch
    knobs.set("BD_SPLIT_SUBRUNS", subruns)
    for L in (100, 127, 150, 300, 383, 400, 441, 500, 513, 529, 540, 639, 700):
        for n_cb, streams in ((64 * subruns * 2 + 37, 2), (64 * subruns, 1), (64 * subruns + 1, 2), (64 * 5 + 63, 1)):
            n = n_cb * L + L // 3 + L                       # + L: the last callback goes to the generic kernel
            x = _synth(bd, streams, n, first=11 + L)
            _compare(bd, oracle, "Spectrum", SR, x, L)
            assert bd.last_kernel_name() == "bd_spectrum_split_kernel"
        n = 64 * 3 * L + 5 * L + 3
        x = _synth(bd, 3, n, first=5)
        flat = torch.zeros(3 + 3 * (n + 1), dtype=torch.int16, device="cuda")
        view = flat[3:].view(3, n + 1)
        view[:, :n] = torch.from_numpy(x).cuda()
        res = bd.analyze_batch(bd.StrategyKind.Spectrum, SR, view, L, n_samples=n)
        ref_v, ref_c, ref_b = util.oracle_batch(oracle, oracle.KIND_SPECTRUM, SR, x, L)
        assert util.values_close(res.values.cpu().numpy(), ref_v, 32768, 3).all(), L
        adjacent = util.threshold_adjacent(ref_v, 2.1e6)
        assert ((res.flags.cpu().numpy() == ref_c) | adjacent).all(), L


@pytest.mark.parametrize("seed", range(6))
def test_spectrum_split_kernel_random_shapes(bd, oracle, seed):
    """Random callback lengths in the split kernel's range, stream counts and lengths (runs of 64
    callbacks cut anywhere), rows at odd addresses with odd strides, random unpinned choices."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    for _ in range(5):
        L = int(rng.integers(65, 767))
        if L < 129 and (L % 32 == 0 or L % 63 == 0):
            L += 1                      # short callbacks on a granule grid keep the granule kernels
        if L % 64 == 0:
            L += 1                      # multiples of 64 have a kernel of their own
        streams = int(rng.integers(1, 7))
        n = int(rng.integers(2 * L, 200 * L))
        kw = {}
        if rng.integers(0, 2):
            kw = [dict(spectrum_include_dc=0), dict(spectrum_divisor=2), dict(spectrum_max_hz=50.0),
                  dict(spectrum_max_hz=10.0), dict(min_amplitude=int(rng.integers(0, 20000)))][int(rng.integers(0, 5))]
        x = _synth(bd, streams, n, first=int(rng.integers(0, 1000)), seed=int(rng.integers(1, 1 << 30)))
        if rng.integers(0, 3) == 0:
            x = rng.integers(-32768, 32768, x.shape).astype(np.int16)        # full-scale noise
        stride = n + int(rng.integers(0, 9))
        shift = int(rng.integers(0, 8))
        flat = torch.zeros(shift + streams * stride, dtype=torch.int16, device="cuda")
        view = flat[shift:].view(streams, stride)
        view[:, :n] = torch.from_numpy(x).cuda()
        bp, op = bd.Params.default(**kw), oracle.default_params(**kw)
        with bd.tuning(split_subruns=int(rng.integers(1, 6)), split_stage=int(rng.integers(0, 3))):
            res = bd.analyze_batch(bd.StrategyKind.Spectrum, SR, view, L, n_samples=n, params=bp)
        assert bd.last_kernel_name() == "bd_spectrum_split_kernel"
        ref_v, ref_c, ref_b = util.oracle_batch(oracle, oracle.KIND_SPECTRUM, SR, x, L, op)
        nb, _, _ = oracle.bin_range(SR, op)
        div = {0: nb, 1: 1024, 2: 0}[op.spectrum_divisor]
        what = (seed, L, streams, n, stride, shift, kw)
        assert util.values_close(res.values.cpu().numpy(), ref_v, 32768, div).all(), what
        adjacent = util.threshold_adjacent(ref_v, op.spectrum_threshold)
        got_c = res.flags.cpu().numpy()
        assert ((got_c == ref_c) | adjacent).all(), what
        if (got_c == ref_c).all():
            for i in range(streams):
                assert res.beats(i) == ref_b[i], what


@pytest.mark.parametrize("L", [63, 126, 441, 504, 693, 819])
def test_spectrum_granules_of_63_samples(bd, oracle, knobs, L):
    """Callback lengths 63 q up to 819 (441 = 10 ms at 44.1 kHz): granules of 63 samples; short streams,
    partial steps, several runs, rows that are only 2-byte aligned, the unpinned choices.  (With three
    bins or fewer the split kernel takes 129 .. 767 samples by default: switched off here.)"""
    import torch
    knobs.set("BD_SPEC_NO_SPLIT", "1")
    for n, streams in ((40000, 3), (4032 * 3 + L * 2 + 17, 2), (max(L, 100), 2), (4032 * 40 + 5, 1), (1024 + L, 2)):
        x = _synth(bd, streams, n, first=3)
        _compare(bd, oracle, "Spectrum", SR, x, L)
        if n // L > 0:
            assert bd.last_kernel_name() == "bd_spectrum_g63_kernel"
    x = _synth(bd, 2, 50000, first=9)
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2),
               dict(spectrum_max_hz=50.0), dict(spectrum_max_hz=10.0)):
        _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**kw), oracle.default_params(**kw))
    n = 30011
    x = _synth(bd, 3, n, first=5)
    flat = torch.zeros(1 + 3 * (n + 3), dtype=torch.int16, device="cuda")
    view = flat[1:].view(3, n + 3)
    view[:, :n] = torch.from_numpy(x).cuda()
    res = bd.analyze_batch(bd.StrategyKind.Spectrum, SR, view, L, n_samples=n)
    ref_v, ref_c, ref_b = util.oracle_batch(oracle, oracle.KIND_SPECTRUM, SR, x, L)
    assert util.values_close(res.values.cpu().numpy(), ref_v, 32768, 3).all()
    np.testing.assert_array_equal(res.flags.cpu().numpy(), ref_c)


@pytest.mark.parametrize("L", [7, 441, 1000, 1500])
def test_spectrum_generic_kernel_alone(bd, oracle, knobs, L):
    """The generic kernel on every callback (what sampling rates with more than three kept bins use)."""
    knobs.set("BD_SPEC_NO_GRANULE", "1")
    x = _synth(bd, 2, 40000, first=3)
    _compare(bd, oracle, "Spectrum", SR, x, L)
    assert bd.last_kernel_name() == "bd_spectrum_generic_kernel"
    x = _synth(bd, 2, 30000, first=4, sr=8000)
    _compare(bd, oracle, "Spectrum", 8000, x, L)
    assert bd.last_kernel_name() == "bd_spectrum_generic_kernel"


@pytest.mark.parametrize("L", [64, 128, 192, 256, 448, 960, 1536, 2048, 3008, 4096,
                               32, 96, 160, 480, 16, 48, 240])
def test_spectrum_granule_callback_lengths(bd, oracle, knobs, L):
    """Multiples of 64 other than 512 / 1024, of 32 up to 512 and of 16 up to 256: the granule kernel
    (granules of 64, 32, 16 samples); short streams, partial steps,
    a stream long enough to be split into runs, and the generic kernel on the same data.  (With three
    bins or fewer the split kernel takes 129 .. 767 samples off the 64-sample grid by default, and the half-block
    kernel the multiples of 512: both switched off here.)"""
    knobs.set("BD_SPEC_NO_SPLIT", "1")
    knobs.set("BD_SPEC_FORCE_GRANULE", "1")
    for n, streams in ((40000, 3), (4096 * 3 + L * 2 + 17, 2), (max(L, 100), 2), (4096 * 40 + 5, 1)):
        x = _synth(bd, streams, n, first=3)
        res, ref_v, _, _ = _compare(bd, oracle, "Spectrum", SR, x, L)
        assert bd.last_kernel_name() == "bd_spectrum_granule_kernel"
    # the unpinned choices and other sampling rates take the same path
    x = _synth(bd, 2, 50000, first=9, sr=22050)
    _compare(bd, oracle, "Spectrum", 22050, x, L)
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2)):
        _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**kw), oracle.default_params(**kw))


def test_spectrum_granule_equals_generic(bd, oracle, knobs):
    import torch
    knobs.set("BD_SPEC_NO_SPLIT", "1")     # 480 and 240 would take the split kernel,
    knobs.set("BD_SPEC_FORCE_GRANULE", "1")    # 2048 the half-block kernel
    x = torch.from_numpy(np.ascontiguousarray(np.pad(_synth(bd, 4, 4096 * 9 + 100, first=1), ((0, 0), (0, 28))))).cuda()
    n = 4096 * 9 + 100
    for L in (256, 2048, 704, 480, 240):
        v1, f1 = bd.batch_candidates(bd.StrategyKind.Spectrum, SR, x, L, n_samples=n)
        knobs.set("BD_SPEC_NO_GRANULE", "1")
        v2, f2 = bd.batch_candidates(bd.StrategyKind.Spectrum, SR, x, L, n_samples=n)
        assert bd.last_kernel_name() == "bd_spectrum_generic_kernel"
        knobs.unset("BD_SPEC_NO_GRANULE")
        assert torch.equal(f1, f2)
        assert torch.isclose(v1, v2, rtol=1e-5, atol=1.0).all()


@pytest.mark.parametrize("sr", [8000, 16000, 22050, 48000, 96000, 192000])
def test_spectrum_other_sampling_rates(bd, oracle, sr):
    x = _synth(bd, 2, 60000, first=5, sr=sr)
    for L in (512, 1024, 600, 529, 300, 700, 270, 200, 150):      # the last five: every flavour of the split kernel above 32 kHz
        _compare(bd, oracle, "Spectrum", sr, x, L)


@pytest.mark.parametrize("sr", [8000, 11025, 12000, 16000, 22050, 24000])
def test_spectrum_more_than_three_bins(bd, oracle, sr):
    """Sampling rates whose band up to 90 Hz holds 4 to 12 bins: callback lengths on the 64-sample
    grid (512 and 1024 included) take the granule kernel with that many bins; the unpinned choices too."""
    x = _synth(bd, 3, 70000, first=5, sr=sr)
    for L in (512, 1024, 256, 2048, 64, 4096):
        _compare(bd, oracle, "Spectrum", sr, x, L)
        assert bd.last_kernel_name() == "bd_spectrum_granule_kernel"
    for L in (450, 1000, 5000, 800):       # off the grids of the granule kernel: the window kernel
        _compare(bd, oracle, "Spectrum", sr, x, L)
        assert bd.last_kernel_name() == "bd_spectrum_window_kernel"
    for L, max_bins in ((160, 8), (480, 8), (96, 8), (240, 5), (48, 5)):   # granules of 32 / 16 samples
        _compare(bd, oracle, "Spectrum", sr, x, L)
        last_bin = oracle.bin_range(sr, oracle.default_params())[2]
        assert bd.last_kernel_name() == ("bd_spectrum_granule_kernel" if last_bin <= max_bins else "bd_spectrum_window_kernel")
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_max_hz=60.0)):
        _compare(bd, oracle, "Spectrum", sr, x, 450, bd.Params.default(**kw), oracle.default_params(**kw))
    for kw in (dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2),
               dict(spectrum_max_hz=60.0)):
        _compare(bd, oracle, "Spectrum", sr, x, 512, bd.Params.default(**kw), oracle.default_params(**kw))
    # below 8 kHz the band holds more bins than the granule kernel evaluates
    _compare(bd, oracle, "Spectrum", 4000, _synth(bd, 2, 30000, first=1, sr=4000), 512)
    assert bd.last_kernel_name() == "bd_spectrum_generic_kernel"


def test_spectrum_unpinned_choices(bd, oracle):
    """Every documented choice for the third-party behaviour is implemented the
    same way on both sides."""
    x = _synth(bd, 2, SR * 4, first=21)
    combos = [
        dict(spectrum_divisor=1), dict(spectrum_divisor=2), dict(spectrum_include_dc=0),
        dict(spectrum_inclusive=0), dict(spectrum_dc_packs_nyquist=1),
        dict(spectrum_divisor=2, spectrum_threshold=3.0e6, spectrum_dc_packs_nyquist=1),
        dict(spectrum_max_hz=43.06640625), dict(spectrum_max_hz=43.06640625, spectrum_inclusive=0),
    ]
    for c in combos:
        for L in (512, 1024, 333):
            _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**c), oracle.default_params(**c))


def test_spectrum_edge_signals(bd, oracle):
    n = 1024 * 24
    rows = [
        np.zeros(n, dtype=np.int16),                                   # silence
        np.full(n, -32768, dtype=np.int16),                            # i16::MIN everywhere (DC = -2^25)
        np.full(n, 32767, dtype=np.int16),
        np.tile(np.array([32767, -32768], dtype=np.int16), n // 2),    # Nyquist tone
        (20000 * np.sin(2 * np.pi * 43.06640625 * np.arange(n) / SR)).astype(np.int16),  # exactly bin 1
        (30000 * np.sin(2 * np.pi * 86.1328125 * np.arange(n) / SR)).astype(np.int16),   # exactly bin 2
        np.where(np.arange(n) % 1024 == 1023, 9829, 0).astype(np.int16),  # just below the amplitude gate
        np.where(np.arange(n) % 1024 == 1023, -9830, 0).astype(np.int16), # exactly at the gate
    ]
    x = np.stack(rows)
    for L in (512, 1024):
        for c in (dict(), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_threshold=1000.0)):
            _compare(bd, oracle, "Spectrum", SR, x, L, bd.Params.default(**c), oracle.default_params(**c))


@pytest.mark.parametrize("L", [1024, 512, 2048, 64, 8])
def test_lpf_aligned(bd, oracle, L):
    x = _synth(bd, 5, SR * 8)
    res, ref_v, ref_c, ref_b = _compare(bd, oracle, "LPF", SR, x, L, exact_values=True)
    assert bd.last_kernel_name() == "bd_lpf_window_kernel"
    if L >= 512:
        assert ref_c.sum() > 0


@pytest.mark.parametrize("L", [1, 3, 63, 65, 100, 127, 129, 160, 257, 317, 441, 480, 529, 882, 1000, 1023, 1025, 1500, 4100])
def test_lpf_ragged(bd, oracle, L):
    """Callback lengths off the 64-sample grid: rows start anywhere inside a 128-byte line; streams of
    many windows, of one, of a partial group of 64, with an odd number of samples after the last."""
    for n, streams in ((50000 + L, 3), (L, 2), (L * 5 + 3, 4), (min(L * 70 + 1, 200000), 1)):
        x = _synth(bd, streams, n, first=9)
        _compare(bd, oracle, "LPF", SR, x, L, exact_values=True)


@pytest.mark.parametrize("L", [64 + 1, 66, 100, 168, 200, 248, 264, 441, 500, 529, 540, 1000, 2001])
def test_lpf_line_staging_layouts(bd, oracle, L):
    """The line-staging kernel (callbacks off the 64-sample grid, 64 samples and longer): stream rows that start at
    every kind of position inside a 128-byte line (odd and even strides, odd base offsets), so that rows begin on a
    line, one sample before or after one, and anywhere between; groups of exactly 64, 65 and 127 windows, one
    window, and garbage after the analysed samples of every stream -- the line a row shares with the next row is
    handed over in registers, the last row's comes from memory and must not look past the stream."""
    torch = _torch()
    rng = np.random.default_rng(L)
    for n_cb, streams in ((64, 3), (65, 2), (127, 2), (1, 3), (200, 2)):
        n = n_cb * L + int(rng.integers(0, L))
        x = _synth(bd, streams, n, first=int(rng.integers(1, 1000)))
        ref_v, ref_c, ref_b = util.oracle_batch(oracle, oracle.KIND_LPF, SR, x, L)
        for base_off, pad in ((0, 0), (1, 3), (63, 64), (64, 1), (5, 70)):
            stride = n + pad
            flat = torch.from_numpy(rng.integers(-32768, 32768, base_off + streams * stride + 64).astype(np.int16)).cuda()
            view = flat[base_off:base_off + streams * stride].view(streams, stride)
            view[:, :n] = torch.from_numpy(x).cuda()
            res = bd.analyze_batch(bd.StrategyKind.LPF, SR, view, L, n_samples=n)
            torch.cuda.synchronize()
            assert bd.last_kernel_name() == "bd_lpf_lines_kernel"
            np.testing.assert_array_equal(res.flags.cpu().numpy(), ref_c)
            np.testing.assert_array_equal(res.values.cpu().numpy(), ref_v)
            for s in range(streams):
                assert res.beats(s) == ref_b[s]


@pytest.mark.parametrize("L,kernel", [(70001, "bd_lpf_lines_kernel"), ((1 << 22) - 1, "bd_lpf_lines_kernel"),
                                      ((1 << 22) + 1, "bd_lpf_window_kernel")])
def test_lpf_very_long_windows(bd, oracle, L, kernel):
    """Windows of tens of thousands to millions of samples: the line-staging kernel addresses a group of windows
    with 32-bit offsets up to 2^22 samples per window; longer ones take the kernel with 64-bit addresses."""
    torch = _torch()
    n = 2 * L + 5
    x = _synth(bd, 2, n, first=31)
    ref_v, ref_c, ref_b = util.oracle_batch(oracle, oracle.KIND_LPF, SR, x, L)
    flat = torch.zeros(3 + 2 * (n + 1), dtype=torch.int16, device="cuda")
    view = flat[3:3 + 2 * (n + 1)].view(2, n + 1)
    view[:, :n] = torch.from_numpy(x).cuda()
    res = bd.analyze_batch(bd.StrategyKind.LPF, SR, view, L, n_samples=n)
    torch.cuda.synchronize()
    assert bd.last_kernel_name() == kernel
    np.testing.assert_array_equal(res.flags.cpu().numpy(), ref_c)
    np.testing.assert_array_equal(res.values.cpu().numpy(), ref_v)
    for s in range(2):
        assert res.beats(s) == ref_b[s]


@pytest.mark.parametrize("L", [8, 16, 24, 32, 48, 64, 96, 104, 128, 192, 1024, 2048])
def test_lpf_several_groups_per_wavefront(bd, oracle, knobs, L):
    """Short windows: a wavefront walks several groups of 64 windows in a row (chosen by the launcher; forced to
    1, 2, 3, 8 and 64 here).  Streams of one window, of a partial group, of whole groups and a partial one, of more
    groups than one wavefront takes; a lowered threshold so that short windows fire too."""
    kw = dict(lpf_ratio=0.05, min_amplitude=100)
    for n, streams in ((L, 2), (L * 37 + 3, 3), (L * 64 * 5 + L * 17 + 1, 2), (L * 64 * 21 + 5, 1)):
        x = _synth(bd, streams, n, first=13)
        for groups in (None, 1, 2, 3, 8, 64):
            if groups is None:
                knobs.unset("BD_LPF_GROUPS")
            else:
                knobs.set("BD_LPF_GROUPS", str(groups))
            _compare(bd, oracle, "LPF", SR, x, L, exact_values=True)
            _, _, ref_c, _ = _compare(bd, oracle, "LPF", SR, x, L, bd.Params.default(**kw), oracle.default_params(**kw),
                                      exact_values=True)
        assert ref_c.sum() > 0 or n == L       # the lowered threshold lets short windows fire
    knobs.unset("BD_LPF_GROUPS")


def test_lpf_many_windows_per_stream(bd, oracle):
    # more than 64 windows per stream, not a multiple of 64; odd stream count
    x = _synth(bd, 7, 1024 * 131 + 77, first=100)
    _compare(bd, oracle, "LPF", SR, x, 1024, exact_values=True)


@pytest.mark.parametrize("sr", [8000, 48000, 65535, 65536, 88200, 96000])
def test_lpf_sampling_rate_wraps_as_u16(bd, oracle, sr):
    # lpf.rs:69 passes `sampling_rate as u16`: 88200 -> 22664, 96000 -> 30464, 65536 -> 0
    x = _synth(bd, 2, 60000, first=2, sr=sr)
    _compare(bd, oracle, "LPF", sr, x, 1024, exact_values=True)


def test_lpf_edge_signals_and_params(bd, oracle):
    n = 1024 * 20
    rows = [
        np.zeros(n, dtype=np.int16),
        np.full(n, -32768, dtype=np.int16),
        np.full(n, 32767, dtype=np.int16),
        np.tile(np.array([32767, -32768], dtype=np.int16), n // 2),
        np.where(np.arange(n) % 1024 < 700, 9830, 0).astype(np.int16),
        np.where(np.arange(n) % 1024 < 700, 9829, 0).astype(np.int16),
        (25000 * np.sin(2 * np.pi * 50 * np.arange(n) / SR)).astype(np.int16),
    ]
    x = np.stack(rows)
    for c in (dict(), dict(lpf_cutoff_hz=4000), dict(lpf_ratio=0.2), dict(lpf_ratio=1.5), dict(lpf_cutoff_hz=0),
              dict(min_amplitude=0, lpf_ratio=0.0)):
        _compare(bd, oracle, "LPF", SR, x, 1024, bd.Params.default(**c), oracle.default_params(**c), exact_values=True)
    # sampling rate 65536 is 0 `as u16` (lpf.rs:69): alpha is NaN, every filtered sample becomes 0, and a window
    # still fires whenever its threshold is <= 0 (a silent window without an amplitude gate, or ratio <= 0)
    for c in (dict(), dict(min_amplitude=0, lpf_ratio=0.0), dict(min_amplitude=0), dict(lpf_ratio=-0.5)):
        for L in (1024, 441):
            _compare(bd, oracle, "LPF", 65536, x, L, bd.Params.default(**c), oracle.default_params(**c),
                     exact_values=True)


def test_refractory_rule_dense_decisions(bd, oracle):
    """All-ones and random decision vectors through bd_refractory against a
    literal replay of the reference's clock + gate."""
    import ctypes as C
    torch = _torch()
    rng = np.random.default_rng(5)
    # the last two: a clock that wraps u32 within the batch (the launcher then leaves every decision to the rule itself)
    for sr, L in ((44100, 1024), (44100, 512), (48000, 480), (8000, 4000), (44100, 1), (96000, 20000), (44100, 100),
                  (44100, 300_000_000), (8000, 49_000_000)):
        n_cb = 700
        flags = np.stack([np.ones(n_cb, np.uint8), np.zeros(n_cb, np.uint8),
                          (rng.random(n_cb) < 0.3).astype(np.uint8), (rng.random(n_cb) < 0.02).astype(np.uint8)])
        Lo = oracle.lib()
        wants = []
        for s in range(flags.shape[0]):
            st = oracle.State()
            Lo.bdo_state_new(C.byref(st), sr)
            want = []
            for j in range(n_cb):
                Lo.bdo_state_update_time(C.byref(st), L)
                if Lo.bdo_last_beat_beyond_threshold(C.byref(st), 400) and flags[s, j]:
                    Lo.bdo_state_mark_beat(C.byref(st))
                    want.append((j, st.beat_time_ms))
            wants.append(want)
        # rows of 700 bytes go a wavefront per stream; rows padded to 704 bytes a lane per stream
        padded = torch.zeros((flags.shape[0], 704), dtype=torch.uint8, device="cuda")
        padded[:, :n_cb] = torch.from_numpy(flags).cuda()
        for rows in (torch.from_numpy(flags).cuda(), padded[:, :n_cb]):
            count, idx, ms = bd.refractory(sr, L, rows, max_beats=n_cb)
            torch.cuda.synchronize()
            count, idx, ms = count.cpu().numpy(), idx.cpu().numpy(), ms.cpu().numpy()
            for s in range(flags.shape[0]):
                got = list(zip(idx[s, :count[s]].tolist(), (ms[s, :count[s]].astype(np.int64) & 0xffffffff).tolist()))
                assert got == wants[s], (sr, L, s, rows.stride(0))


@pytest.mark.parametrize("n_cb", [1, 15, 16, 17, 31, 1023, 1024, 1025, 1039, 2049, 4095, 4096, 4097, 5119, 5120, 5167, 8192, 8193,
                                  20671, 24576, 24577])
def test_refractory_rows_on_16_byte_boundaries_and_any_nonzero_byte(bd, oracle, n_cb):
    """The kernels of the refractory pass against the reference's clock and gate.  Rows on 16-byte boundaries of 16 to
    24,576 callbacks: 8 streams per wavefront, packed to bits in segments of 4,096 callbacks, a lane per stream; other
    rows: a wavefront per stream, 1,024 decisions per step on long rows on 16-byte boundaries, 64 otherwise.  13 streams
    (a full wavefront of 8 and one of 5); a decision is any non-zero byte (0x01, 0x80, 0xff, ...); bytes between n_cb and
    the row stride are not decisions."""
    import ctypes as C
    torch = _torch()
    rng = np.random.default_rng(n_cb)
    sr, L = 44100, 512
    n_streams = 13
    stride = ((n_cb + 15) & ~15) + 16
    host = rng.integers(1, 256, (n_streams, stride)).astype(np.uint8)        # garbage past n_cb
    dens = [1.0, 0.0, 0.5, 0.05, 0.01, 0.2, 0.002, 0.1, 0.0005, 1.0, 0.03, 0.3, 0.007]
    for s in range(n_streams):
        host[s, :n_cb] = np.where(rng.random(n_cb) < dens[s], rng.choice([1, 2, 0x80, 0xff], n_cb), 0)
    Lo = oracle.lib()
    wants = []
    for s in range(n_streams):
        st = oracle.State()
        Lo.bdo_state_new(C.byref(st), sr)
        want = []
        for j in range(n_cb):
            Lo.bdo_state_update_time(C.byref(st), L)
            if host[s, j] and Lo.bdo_last_beat_beyond_threshold(C.byref(st), 400):
                Lo.bdo_state_mark_beat(C.byref(st))
                want.append((j, st.beat_time_ms))
        wants.append(want)
    # rows on 16-byte boundaries: the launcher's choice, and (switch) a wavefront per stream; rows off them: a wavefront per stream
    for shift, wave in ((0, 0), (0, 1), (3, 0)):
        flat = torch.zeros(shift + host.size + 16, dtype=torch.uint8, device="cuda")
        view = flat[shift:shift + host.size].view(n_streams, stride)
        view.copy_(torch.from_numpy(host).cuda())
        with bd.tuning(refractory_wave=wave or None):
            count, idx, ms = bd.refractory(sr, L, view[:, :n_cb], max_beats=n_cb)
            torch.cuda.synchronize()
        count, idx, ms = count.cpu().numpy(), idx.cpu().numpy(), ms.cpu().numpy()
        for s in range(n_streams):
            got = list(zip(idx[s, :count[s]].tolist(), ms[s, :count[s]].tolist()))
            assert got == wants[s], (n_cb, shift, wave, s)


def test_beat_list_capacity_is_reported_not_overrun(bd):
    torch = _torch()
    flags = torch.ones((3, 500), dtype=torch.uint8, device="cuda")
    count, idx, ms = bd.refractory(44100, 1024, flags, max_beats=4)
    torch.cuda.synchronize()
    assert (count.cpu().numpy() > 4).all()      # true number of beats
    assert idx.shape == (3, 4)


@pytest.mark.parametrize("kind_name", ["LPF", "Spectrum"])
@pytest.mark.parametrize("L", [1024, 512, 529])
def test_detector_handle_matches_reference_call_by_call(bd, oracle, kind_name, L):
    """StrategyKind::detector + Strategy::is_beat, callback by callback."""
    kind = bd.StrategyKind[kind_name]
    okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
    x = _synth(bd, 1, SR * 5, first=4)[0]
    det = kind.detector(SR)
    ref = oracle.OracleDetector(okind, SR)
    assert det.kind() == kind
    n_beats = 0
    for j in range(len(x) // L):
        w = x[j * L:(j + 1) * L]
        got = det.is_beat(w)
        want = ref.is_beat(w)
        assert (got.relative_ms() if got else None) == want, f"callback {j}"
        n_beats += want is not None
    assert n_beats > 0
    det.close()


def test_detector_handle_variable_callback_lengths(bd, oracle):
    # Linux callbacks: "mostly 500-540" samples (spectrum.rs:40-42)
    rng = np.random.default_rng(3)
    x = _synth(bd, 1, SR * 6, first=8)[0]
    for kind_name in ("LPF", "Spectrum"):
        kind = bd.StrategyKind[kind_name]
        okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
        det, ref = kind.detector(SR), oracle.OracleDetector(okind, SR)
        pos = 0
        while pos < len(x) - 3000:
            n = int(rng.choice([500, 512, 529, 540, 1, 1024, 2500]))
            w = x[pos:pos + n]
            got, want = det.is_beat(w), ref.is_beat(w)
            assert (got.relative_ms() if got else None) == want
            pos += n
        det.close()


def test_detector_empty_window_is_an_error(bd):
    det = bd.StrategyKind.Spectrum.detector(SR)
    with pytest.raises(bd.BeatDetectorError) as e:
        det.is_beat(np.zeros(0, dtype=np.int16))
    assert e.value.status == -5
    det.close()


def test_converters_match_reference(bd, oracle):
    torch = _torch()
    u = np.arange(65536, dtype=np.uint16)
    got = bd.u16_data_to_i16(torch.from_numpy(u.view(np.int16)).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.u16_data_to_i16(u))
    rng = np.random.default_rng(0)
    f = np.concatenate([rng.uniform(-1.2, 1.2, 100000).astype(np.float32),
                        np.array([0, 1, -1, np.nan, np.inf, -np.inf, 1e-8, -1e-8, 0.99999, 3.0e9], dtype=np.float32)])
    got = bd.f32_data_to_i16(torch.from_numpy(f).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.f32_data_to_i16(f))


def test_stereo_downmix_matches_reference(bd, oracle):
    torch = _torch()
    rng = np.random.default_rng(5)
    st = rng.integers(-32768, 32768, (3, 2 * 5001)).astype(np.int16)
    st[0, :8] = [-32768, -32768, 32767, 32767, -1, 0, 1, -2]          # extremes and odd sums of either sign
    got = bd.stereo_to_mono(torch.from_numpy(st).cuda()).cpu().numpy()
    for r in range(3):
        np.testing.assert_array_equal(got[r], oracle.stereo_to_mono(st[r]))


@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000, 4099])
@pytest.mark.parametrize("shift", [0, 1, 3])
def test_converters_lengths_and_alignments(bd, oracle, n, shift):
    """Whole 16-byte units, tails of 1 .. 7 samples, and buffers that start off a 16-byte boundary."""
    torch = _torch()
    rng = np.random.default_rng(n * 7 + shift)
    u = rng.integers(0, 65536, n + shift).astype(np.uint16)
    got = bd.u16_data_to_i16(torch.from_numpy(u.view(np.int16)).cuda()[shift:]).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.u16_data_to_i16(u[shift:]))
    f = rng.uniform(-1.3, 1.3, n + shift).astype(np.float32)
    f[::5] = np.nan if n > 20 else f[::5]
    got = bd.f32_data_to_i16(torch.from_numpy(f).cuda()[shift:]).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.f32_data_to_i16(f[shift:]))
    st = rng.integers(-32768, 32768, 2 * (n + shift)).astype(np.int16)
    got = bd.stereo_to_mono(torch.from_numpy(st).cuda()[2 * shift:]).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.stereo_to_mono(st[2 * shift:]))


def test_generator_device_equals_host(bd):
    torch = _torch()
    for n_streams, n, first in ((5, 44100 * 3 + 5, 0), (3, 1000, 77), (2, 8, 5), (1, 7, 2)):
        stride = (n + 7) & ~7
        buf = torch.zeros((n_streams, stride), dtype=torch.int16, device="cuda")
        bd.synth_fill(buf, n_samples=n, first_stream=first, seed=1234)
        torch.cuda.synchronize()
        host = bd.synth_fill_host(n_streams, n, first_stream=first, seed=1234)
        np.testing.assert_array_equal(buf[:, :n].cpu().numpy(), host)


def test_host_pointer_entry_points(bd, oracle):
    x = _synth(bd, 9, SR * 3, first=30)
    for kind_name in ("LPF", "Spectrum"):
        kind = bd.StrategyKind[kind_name]
        okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
        _, ref_c, ref_b = util.oracle_batch(oracle, okind, SR, x, 1024)
        # (a device listed three times = three contiguous stream shards with a host thread each)
        for res in (bd.analyze_batch_host(kind, SR, x, 1024), bd.analyze_batch_multi(kind, SR, x, 1024, devices=[0]),
                    bd.analyze_batch_multi(kind, SR, x, 1024, devices=[0, 0, 0]), bd.analyze_batch_multi(kind, SR, x, 441, devices=[0, 0])):
            if res.flags.shape != ref_c.shape:      # the 441-sample case has its own reference
                _, c441, b441 = util.oracle_batch(oracle, okind, SR, x, 441)
                np.testing.assert_array_equal(res.flags, c441)
                assert all(res.beats(s) == b441[s] for s in range(x.shape[0]))
                continue
            np.testing.assert_array_equal(res.flags, ref_c)
            for s in range(x.shape[0]):
                assert res.beats(s) == ref_b[s]


def test_resident_shards_on_listed_devices(bd, oracle):
    """bd_analyze_batch_devices: shards that already live on their GPU, one host thread each inside the library.
    (One GPU here: it is listed for every shard.)  Shards of different sizes, an empty one, both strategies, a
    callback length off the grid; every shard equals the same streams analysed in one batch and the oracle."""
    torch = _torch()
    x = _synth(bd, 11, SR * 3 + 77, first=30)
    n = x.shape[1]
    dev = torch.zeros((11, (n + 63) & ~63), dtype=torch.int16, device="cuda")
    dev[:, :n] = torch.from_numpy(x).cuda()
    cuts = [0, 4, 4, 9, 11]                                  # shards of 4, 0, 5 and 2 streams
    shards = [dev[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    for kind_name, L in (("Spectrum", 512), ("LPF", 1024), ("Spectrum", 529), ("LPF", 441)):
        kind = bd.StrategyKind[kind_name]
        okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
        _, ref_c, ref_b = util.oracle_batch(oracle, okind, SR, x, L)
        whole = bd.analyze_batch(kind, SR, dev, L, n_samples=n)
        outs = bd.analyze_batch_devices(kind, SR, shards, L, n_samples=n)
        assert len(outs) == 4 and outs[1].flags.shape[0] == 0
        for (a, b), res in zip(zip(cuts[:-1], cuts[1:]), outs):
            np.testing.assert_array_equal(res.flags.cpu().numpy(), ref_c[a:b])
            assert torch.equal(res.values, whole.values[a:b]) and torch.equal(res.beat_count, whole.beat_count[a:b])
            for s in range(b - a):
                assert res.beats(s) == ref_b[a + s]
        no_values = bd.analyze_batch_devices(kind, SR, shards, L, n_samples=n, want_values=False)
        assert all(r.values is None for r in no_values)
        assert all(torch.equal(r.flags, o.flags) and torch.equal(r.beat_idx, o.beat_idx) for r, o in zip(no_values, outs))
    with pytest.raises(bd.BeatDetectorError):
        bd.analyze_batch_devices(bd.StrategyKind.LPF, SR, [dev[:2], dev[2:5, :1024]], 1024)   # strides differ


def test_host_entry_in_many_chunks(bd, oracle, knobs):
    """The host-pointer entry with chunks of 3 streams (two slots reused many times), all outputs requested."""
    n = SR * 2 + 100
    x = np.zeros((20, n + 37), dtype=np.int16)
    x[:, :n] = _synth(bd, 20, n, first=200)
    knobs.set("BD_HOST_CHUNK_BYTES", str(3 * ((n + 63) & ~63) * 2))
    for kind, okind in ((bd.StrategyKind.Spectrum, oracle.KIND_SPECTRUM), (bd.StrategyKind.LPF, oracle.KIND_LPF)):
        _, ref_c, ref_b = util.oracle_batch(oracle, okind, SR, x[:, :n], 512)
        res = bd.analyze_batch_host(kind, SR, x[:, :n], 512)
        np.testing.assert_array_equal(res.flags, ref_c)
        for s in range(20):
            assert res.beats(s) == ref_b[s]


def test_full_size_properties_spectrum(bd):
    """At a size the oracle cannot finish: size-independent properties.
    (a) a stream's results do not depend on which batch it is analysed in;
    (b) decisions are invariant under sign inversion of the samples (|X| and
        max|x| are even), values agree within tolerance;
    (c) hop-512 callbacks with odd index see the same ring as 1024-callbacks:
        values[2j+1] at L=512 equals values[j] at L=1024."""
    torch = _torch()
    n_streams, n = 512, SR * 60
    stride = (n + 7) & ~7
    buf = torch.zeros((n_streams, stride), dtype=torch.int16, device="cuda")
    bd.synth_fill(buf, n_samples=n, seed=99)
    K = bd.StrategyKind.Spectrum
    full = bd.analyze_batch(K, SR, buf, 512, n_samples=n)
    sub = bd.analyze_batch(K, SR, buf[100:117].contiguous(), 512, n_samples=n)
    torch.cuda.synchronize()
    assert torch.equal(full.flags[100:117], sub.flags)
    assert torch.equal(full.beat_count[100:117], sub.beat_count)
    assert torch.equal(full.beat_ms[100:117], sub.beat_ms)
    neg = bd.analyze_batch(K, SR, torch.clamp(-buf.to(torch.int32), -32768, 32767).to(torch.int16), 512, n_samples=n)
    torch.cuda.synchronize()
    close = torch.isclose(neg.values, full.values, rtol=1e-5, atol=1.0)
    assert close.all()
    assert (neg.flags != full.flags).sum().item() <= 2
    l1024 = bd.analyze_batch(K, SR, buf, 1024, n_samples=n)
    torch.cuda.synchronize()
    n2 = l1024.values.shape[1]
    assert torch.isclose(full.values[:, 1:2 * n2:2], l1024.values, rtol=1e-5, atol=1.0).all()
    assert int(full.beat_count.sum()) > 0


def test_rows_that_are_only_2_byte_aligned(bd, oracle):
    """Odd strides and an odd base offset: every strategy and callback length still matches."""
    torch = _torch()
    n = SR * 2 + 11
    x = _synth(bd, 4, n, first=40)
    stride = n + 3
    flat = torch.zeros(1 + 4 * stride, dtype=torch.int16, device="cuda")
    view = flat[1:].view(4, stride)
    view[:, :n] = torch.from_numpy(x).cuda()
    assert view.data_ptr() % 4 == 2
    for kind_name, L, kernel in (("Spectrum", 512, "bd_spectrum_granule_kernel"), ("Spectrum", 1024, "bd_spectrum_granule_kernel"),
                                 ("Spectrum", 2048, "bd_spectrum_granule_kernel"), ("Spectrum", 3072, "bd_spectrum_granule_kernel"), ("Spectrum", 441, "bd_spectrum_split_kernel"),
                                 ("Spectrum", 819, "bd_spectrum_g63_kernel"),
                                 ("LPF", 1024, "bd_lpf_lines_kernel"), ("LPF", 441, "bd_lpf_lines_kernel"),
                                 ("LPF", 50, "bd_lpf_window_kernel")):
        kind = bd.StrategyKind[kind_name]
        okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
        res = bd.analyze_batch(kind, SR, view, L, n_samples=n)
        torch.cuda.synchronize()
        assert bd.last_kernel_name() == kernel
        ref_v, ref_c, ref_b = util.oracle_batch(oracle, okind, SR, x, L)
        np.testing.assert_array_equal(res.flags.cpu().numpy(), ref_c)
        for s in range(4):
            assert res.beats(s) == ref_b[s]
        if kind_name == "LPF":
            np.testing.assert_array_equal(res.values.cpu().numpy(), ref_v)
        else:
            assert util.values_close(res.values.cpu().numpy(), ref_v, 32768, 3).all()


@pytest.mark.parametrize("kind_name", ["Spectrum", "LPF"])
def test_ragged_batch(bd, oracle, kind_name):
    """Streams of different lengths in one batch (zero, shorter than a callback, exact multiples,
    trailing partial callbacks; garbage after each stream's own samples): every stream as if alone."""
    torch = _torch()
    kind = bd.StrategyKind[kind_name]
    okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
    lens = [0, 100, 1024, 1025, SR * 3, SR * 3 + 511, 512 * 40, SR * 5 - 1]
    nmax = max(lens)
    x = _synth(bd, len(lens), nmax, first=11)
    rng = np.random.default_rng(1)
    host = x.copy()
    for i, n in enumerate(lens):
        host[i, n:] = rng.integers(-32768, 32768, nmax - n)      # what follows a stream must not matter
    buf = torch.zeros((len(lens), (nmax + 63) & ~63), dtype=torch.int16, device="cuda")
    buf[:, :nmax] = torch.from_numpy(host).cuda()
    lengths = torch.tensor(lens, dtype=torch.int64, device="cuda")
    for L in (512, 1024, 256, 441, 450, 480):
        res = bd.analyze_batch_ragged(kind, SR, buf, lengths, L)
        torch.cuda.synchronize()
        for i, n in enumerate(lens):
            v, c, beats = oracle.run_stream(okind, SR, x[i, :n], L, full=True) if n >= L else (np.zeros(0), np.zeros(0), [])
            assert res.beats(i) == beats, (L, i)
            assert int(res.beat_count[i]) == len(beats)
            np.testing.assert_array_equal(res.flags[i, :len(c)].cpu().numpy(), c)


def test_tuning_knobs_do_not_change_results(bd, oracle, knobs):
    """Every value of the experiment knobs selects code that must give the same answers."""
    x = _synth(bd, 4, SR * 10, first=60)
    for run in ("8", "16", "64", "256", "512"):
        knobs.set("BD_SPEC_RUN", run)
        for L in (512, 1024):
            _compare(bd, oracle, "Spectrum", SR, x, L)
    knobs.unset("BD_SPEC_RUN")
    for variant in ("1", "2", "3"):
        knobs.set("BD_LPF_VARIANT", variant)
        for L in (1024, 64, 4096, 136):
            _compare(bd, oracle, "LPF", SR, x, L, exact_values=True)
    for variant in ("5", "6"):              # line staging: every line from memory / per-piece addressing
        knobs.set("BD_LPF_VARIANT", variant)
        for L in (441, 65, 200, 1000):
            _compare(bd, oracle, "LPF", SR, x, L, exact_values=True)
    knobs.set("BD_LPF_VARIANT", "7")        # line staging for the lengths of the aligned staging as well
    for L in (1024, 64, 4096, 136, 2048, 192):
        _compare(bd, oracle, "LPF", SR, x, L, exact_values=True)
        assert bd.last_kernel_name() == "bd_lpf_lines_kernel"
    knobs.unset("BD_LPF_VARIANT")
    knobs.set("BD_SPEC_NO_QUARTER", "1")    # 256- and 128-sample callbacks through the granule kernel
    for L in (256, 128):
        _compare(bd, oracle, "Spectrum", SR, x, L)
        assert bd.last_kernel_name() == "bd_spectrum_granule_kernel"
    knobs.unset("BD_SPEC_NO_QUARTER")
    knobs.set("BD_REFRACTORY_WAVE", "1")    # the refractory pass with a wavefront per stream on every row
    for kind_name, L in (("Spectrum", 512), ("LPF", 1024), ("Spectrum", 64), ("LPF", 16)):
        _compare(bd, oracle, kind_name, SR, x, L, exact_values=(kind_name == "LPF"))
    knobs.unset("BD_REFRACTORY_WAVE")


def test_row_layouts(bd, oracle):
    """Misaligned stream rows (16-byte aligned but not on 128-byte lines), dense and
    padded output rows: same results."""
    torch = _torch()
    n = SR * 4
    x = _synth(bd, 5, n, first=70)
    for stride in (n, n + 8, n + 24, n + 40, ((n + 63) & ~63)):      # n = 176400 is a multiple of 16 samples
        buf = torch.zeros((5, stride), dtype=torch.int16, device="cuda")
        buf[:, :n] = torch.from_numpy(x).cuda()
        for kind_name, L in (("LPF", 1024), ("LPF", 512), ("LPF", 64), ("Spectrum", 512), ("Spectrum", 1024)):
            kind = bd.StrategyKind[kind_name]
            okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
            n_cb = n // L
            dense_v = torch.zeros((5, n_cb), dtype=torch.float32, device="cuda")
            dense_f = torch.zeros((5, n_cb), dtype=torch.uint8, device="cuda")
            bd.batch_candidates(kind, SR, buf, L, n_samples=n, values=dense_v, flags=dense_f)
            pv, pf = bd.batch_candidates(kind, SR, buf, L, n_samples=n)
            torch.cuda.synchronize()
            assert torch.equal(dense_f, pf) and torch.equal(dense_v, pv)
            _, ref_c, _ = util.oracle_batch(oracle, okind, SR, x, L)
            np.testing.assert_array_equal(pf.cpu().numpy(), ref_c)
    # a view that starts in the middle of a row (odd 16-byte offset)
    buf = torch.zeros((5, n + 64), dtype=torch.int16, device="cuda")
    buf[:, 8:8 + n] = torch.from_numpy(x).cuda()
    view = buf[:, 8:]
    assert view.data_ptr() % 128 == 16
    L_ = bd.lib()
    flags = torch.zeros((5, n // 1024), dtype=torch.uint8, device="cuda")
    for kind_name in ("LPF", "Spectrum"):
        okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
        rc = L_.bd_batch_candidates(bd.StrategyKind[kind_name].value, SR, None, view.data_ptr(), 5, n, n + 64, 1024,
                                    None, flags.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        _, ref_c, _ = util.oracle_batch(oracle, okind, SR, x, 1024)
        np.testing.assert_array_equal(flags.cpu().numpy(), ref_c)


def test_concurrent_callers_on_their_own_streams(bd, oracle):
    """Several host threads, each enqueueing batches on a HIP stream of its own (ctypes releases the GIL);
    also the host-pointer entry from two threads at once.  Every result equals the single-threaded one."""
    import threading
    torch = _torch()
    jobs = [("Spectrum", 512), ("LPF", 1024), ("Spectrum", 441), ("LPF", 441), ("Spectrum", 1000), ("Spectrum", 160)]
    x = _synth(bd, 6, SR * 4 + 123, first=60)
    refs = {j: util.oracle_batch(oracle, oracle.KIND_SPECTRUM if j[0] == "Spectrum" else oracle.KIND_LPF, SR, x, j[1]) for j in jobs}
    dev = torch.zeros((6, (x.shape[1] + 7) & ~7), dtype=torch.int16, device="cuda")
    dev[:, :x.shape[1]] = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    errors = []

    def device_worker(job):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(5):
                    res = bd.analyze_batch(bd.StrategyKind[job[0]], SR, dev, job[1], n_samples=x.shape[1], stream=st)
                st.synchronize()
            ref_v, ref_c, ref_b = refs[job]
            np.testing.assert_array_equal(res.flags.cpu().numpy(), ref_c)
            assert all(res.beats(s) == ref_b[s] for s in range(6))
        except Exception as e:       # surfaced in the main thread
            errors.append((job, repr(e)))

    def host_worker(job):
        try:
            res = bd.analyze_batch_host(bd.StrategyKind[job[0]], SR, x, job[1])
            np.testing.assert_array_equal(res.flags, refs[job][1])
        except Exception as e:
            errors.append((job, repr(e)))

    threads = [threading.Thread(target=device_worker, args=(j,)) for j in jobs]
    threads += [threading.Thread(target=host_worker, args=(j,)) for j in jobs[:2]]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
