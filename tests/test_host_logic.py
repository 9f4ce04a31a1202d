"""Host-side logic of the library that needs no GPU: parameters, strategy
metadata, sizing helpers, the integer generator, error behaviour."""
import numpy as np
import pytest


def test_default_params_equal_the_reference_constants(bd, oracle):
    p, q = bd.Params.default(), oracle.default_params()
    for name, _ in p._fields_:
        assert getattr(p, name) == getattr(q, name), name
    assert p.min_amplitude == 9830 and p.min_beat_gap_ms == 400


def test_strategy_kind_metadata(bd):
    # src/lib.rs:142-185, spectrum.rs:106-131, lpf.rs:89-114
    K = bd.StrategyKind
    assert K.values() == [K.LPF, K.Spectrum]
    assert (K.LPF.value, K.Spectrum.value) == (0, 1)
    assert K.LPF.name_() == "Simple Lowpass Filter"
    assert K.Spectrum.name_() == "Simple Spectrum Analysis"
    assert K.LPF.description().startswith("A simple beat detection using a lowpass filter.")
    assert K.Spectrum.description().startswith("A simple beat detection using a spectrum analysis.")
    assert K.LPF.description().endswith("pretty good results in 'easy' music, like most of 90s pop hits.")
    assert K.LPF.min_duration_between_beats_ms() == 400
    assert K.Spectrum.min_duration_between_beats_ms() == 400
    L = bd.lib()
    assert L.bd_kind_count() == 2
    assert L.bd_kind_name(7) is None and L.bd_kind_description(-1) is None


def test_beat_info(bd):
    b = bd.BeatInfo(402)
    assert b.relative_ms() == 402 and b == bd.BeatInfo(402) and "402" in repr(b)


def test_batch_sizing_helpers(bd):
    assert bd.batch_callbacks(2646000, 1024) == 2583      # src/lib.rs:276 floor
    assert bd.batch_callbacks(2646000, 512) == 5167
    assert bd.batch_callbacks(5, 0) == 0
    # beats are at least ceil(400/23) = 18 callbacks apart at 1024 samples / 44.1 kHz
    mb = bd.batch_max_beats(bd.StrategyKind.Spectrum, 44100, 2646000, 1024)
    assert 2583 // 18 <= mb <= 2583 // 18 + 2
    mb = bd.batch_max_beats(bd.StrategyKind.LPF, 44100, 2646000, 512)
    assert 5167 // 37 <= mb <= 5167 // 37 + 2            # ceil(400/11) = 37
    assert bd.batch_max_beats(bd.StrategyKind.LPF, 44100, 100, 1) == 100  # step 0 ms: every callback may fire


def test_generator_is_deterministic_and_batch_independent(bd):
    a = bd.synth_fill_host(4, 50000, first_stream=10, seed=99)
    b = bd.synth_fill_host(2, 50000, first_stream=12, seed=99)
    np.testing.assert_array_equal(a[2:], b)                # a stream does not depend on its batch
    c = bd.synth_fill_host(4, 50000, first_stream=10, seed=100)
    assert (a != c).any()
    assert np.abs(a.astype(np.int32)).max() > 9830         # clicks pass the amplitude gate
    assert np.abs(a.astype(np.int32)).max() <= 32768
    # checksums of the generator's output (a change here invalidates the golden fixtures)
    assert int(a.astype(np.int64).sum()) == 22691556
    assert int(np.bitwise_xor.reduce(a.astype(np.int64).ravel() * np.arange(1, a.size + 1))) == 841698703


def test_invalid_arguments_are_reported(bd):
    L = bd.lib()
    import ctypes as C
    h = C.c_void_p()
    assert L.bd_detector_create(5, 44100, None, -1, C.byref(h)) == -1          # unknown kind
    assert L.bd_detector_create(1, 0, None, -1, C.byref(h)) == -1              # zero sampling rate
    assert L.bd_detector_create(1, 100, None, -1, C.byref(h)) == -6            # 90 Hz above Nyquist: reference panics
    p = bd.Params.default(spectrum_include_dc=0)
    assert L.bd_detector_create(1, 96000, C.byref(p), -1, C.byref(h)) == -6    # only bin 0 <= 90 Hz and it is excluded
    assert b"bin" in L.bd_last_error()
    assert L.bd_strerror(-6).decode().startswith("no spectrum bin")
    x = np.zeros(16, dtype=np.int16)
    assert L.bd_synth_fill_host(x.ctypes.data_as(C.c_void_p), 1, 16, 8, 0, 0, 44100) == -1  # stride < n


def test_no_gpu_means_an_error_not_a_cpu_result(bd):
    """There is no CPU path: without a GPU, compute entry points fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bd.BeatDetectorError) as e:
        bd.StrategyKind.Spectrum.detector(44100)
    assert e.value.status == -2
    with pytest.raises(bd.BeatDetectorError) as e:
        bd.analyze_batch_host(bd.StrategyKind.LPF, 44100, np.zeros((1, 4096), dtype=np.int16), 1024)
    assert e.value.status == -2
    assert bd.device_count() == 0


def test_product_package_does_not_touch_the_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "beat_detector_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"\boracle\b|bdo_|bd_oracle", text), os.path.join(dirpath, f)


def test_multi_device_entry_validates_before_it_starts_workers(bd):
    """Argument errors of bd_analyze_batch_multi come back once, before any pointer is offset per shard."""
    import ctypes as C
    L = bd.lib()
    dev = (C.c_int * 2)(0, 0)
    x = (C.c_int16 * 4096)()
    cnt = (C.c_uint32 * 2)()
    vp = lambda a: C.cast(a, C.c_void_p)
    assert L.bd_analyze_batch_multi(1, 44100, None, None, 2, 2048, 2048, 512, None, None, vp(cnt), None, None, 0, dev, 2) == -1
    assert L.bd_analyze_batch_multi(1, 44100, None, vp(x), 2, 2048, 2048, 512, None, None, None, None, None, 0, dev, 2) == -1
    assert L.bd_analyze_batch_multi(1, 44100, None, vp(x), 2, 2048, 1024, 512, None, None, vp(cnt), None, None, 0, dev, 2) == -1
    assert L.bd_analyze_batch_multi(1, 44100, None, vp(x), 2, 2048, 2048, 0, None, None, vp(cnt), None, None, 0, dev, 2) == -5
    assert L.bd_analyze_batch_multi(7, 44100, None, vp(x), 2, 2048, 2048, 512, None, None, vp(cnt), None, None, 0, dev, 2) == -1
    assert L.bd_analyze_batch_multi(1, 44100, None, vp(x), 2, 2048, 2048, 512, None, None, vp(cnt), None, None, 0, None, 0) == -1
    assert L.bd_analyze_batch_multi(1, 44100, None, vp(x), 0, 2048, 2048, 512, None, None, vp(cnt), None, None, 0, dev, 2) == 0


def test_resident_shards_entry_validates_before_it_starts_workers(bd):
    """bd_analyze_batch_devices: missing arrays, bad kinds, lengths and strides are reported once, up front (no
    pointer of a shard is touched); without a GPU a well-formed call says so instead of computing on the CPU."""
    import ctypes as C
    import torch
    L = bd.lib()
    vp, u64 = C.c_void_p, C.c_uint64
    dev = (C.c_int * 2)(0, 0)
    ptrs = (vp * 2)(0x1000, 0x2000)          # never dereferenced by the checks
    n = (u64 * 2)(3, 4)
    none = (vp * 2)(None, None)
    call = lambda kind, samples, n_streams, n_samples, stride, cb, flags, count, devices, n_dev: L.bd_analyze_batch_devices(
        kind, 44100, None, samples, n_streams, n_samples, stride, cb, None, flags, 0, count, None, None, 0, devices, n_dev)
    assert call(1, ptrs, n, 2048, 2048, 512, ptrs, ptrs, None, 0) == -1        # no devices
    assert call(1, None, n, 2048, 2048, 512, ptrs, ptrs, dev, 2) == -1         # no sample array
    assert call(1, ptrs, None, 2048, 2048, 512, ptrs, ptrs, dev, 2) == -1      # no stream counts
    assert call(1, ptrs, n, 2048, 2048, 512, None, ptrs, dev, 2) == -1         # no decision rows
    assert call(1, ptrs, n, 2048, 2048, 512, ptrs, None, dev, 2) == -1         # no beat counts
    assert call(7, ptrs, n, 2048, 2048, 512, ptrs, ptrs, dev, 2) == -1         # unknown strategy
    assert call(1, ptrs, n, 2048, 2048, 0, ptrs, ptrs, dev, 2) == -5           # empty callbacks
    assert call(1, ptrs, n, 2048, 1024, 512, ptrs, ptrs, dev, 2) == -1         # stride shorter than the stream
    if not torch.cuda.is_available():
        assert call(1, ptrs, n, 2048, 2048, 512, ptrs, ptrs, dev, 2) == -2     # BD_ERR_NO_DEVICE
    else:
        assert call(1, none, n, 2048, 2048, 512, ptrs, ptrs, dev, 2) == -1     # a shard without samples
        bad = (C.c_int * 2)(0, 1 << 20)
        assert call(1, ptrs, n, 2048, 2048, 512, ptrs, ptrs, bad, 2) == -1     # device out of range
