"""The oracle against everything the reference itself pins: its one
known-answer test (the millisecond clock), its source constants and the
literal behaviour of its integer helpers."""
import numpy as np
import pytest


def test_clock_known_answer(oracle):
    # src/strategies/mod.rs:122-164 test_analysis_state_get_relative_time_ms
    import ctypes as C
    s = oracle.State()
    L = oracle.lib()
    L.bdo_state_new(C.byref(s), 44100)
    L.bdo_state_update_time(C.byref(s), 1024)
    assert (s.beat_time_ms, s.time_ms) == (23 // 2, 23)
    L.bdo_state_update_time(C.byref(s), 1024)
    assert (s.beat_time_ms, s.time_ms) == (int(23.0 * 1.5), 46)
    L.bdo_state_update_time(C.byref(s), 317)
    assert s.beat_time_ms == 46 + int(317.0 / 2.0 * 1.0 / 44100.0 * 1000.0)
    assert s.time_ms == 46 + int(317.0 / 1.0 / 44100.0 * 1000.0)
    assert (s.beat_time_ms, s.time_ms) == (49, 53)


def test_clock_matches_float32_replay(oracle):
    import ctypes as C
    L = oracle.lib()
    for sr in (8000, 22050, 44100, 48000, 96000):
        s = oracle.State()
        L.bdo_state_new(C.byref(s), sr)
        mps = np.float32(np.float32(1.0) / np.float32(sr)) * np.float32(1000.0)
        assert np.float32(s.ms_per_sample) == mps
        t = 0
        for n in (1024, 512, 317, 1, 529, 2048, 1024):
            L.bdo_state_update_time(C.byref(s), n)
            ms = np.float32(mps * np.float32(n))
            assert s.beat_time_ms == t + int(np.float32(ms / np.float32(2.0)))
            t += int(ms)
            assert s.time_ms == t


def test_default_constants(oracle):
    p = oracle.default_params()
    assert p.min_amplitude == 9830            # src/lib.rs:134
    assert p.min_beat_gap_ms == 400           # spectrum.rs:130, lpf.rs:113
    assert p.spectrum_max_hz == 90.0          # spectrum.rs:85
    assert p.spectrum_threshold == 2_100_000.0  # spectrum.rs:92
    assert p.lpf_cutoff_hz == 120             # lpf.rs:65
    assert np.float32(p.lpf_ratio) == np.float32(0.77)  # lpf.rs:74


def test_refractory_gate(oracle):
    # src/lib.rs:115-127
    import ctypes as C
    L = oracle.lib()
    s = oracle.State()
    L.bdo_state_new(C.byref(s), 44100)
    assert L.bdo_last_beat_beyond_threshold(C.byref(s), 400) == 1  # beat_time == 0
    n_first = None
    for k in range(40):
        L.bdo_state_update_time(C.byref(s), 1024)
        ok = L.bdo_last_beat_beyond_threshold(C.byref(s), 400)
        assert s.beat_time_ms == 23 * k + 11
        assert ok == (1 if s.beat_time_ms >= 400 else 0)
        if ok and n_first is None:
            n_first = k
    assert n_first == 17  # 17*23+11 = 402: nothing can fire before 400 ms


def test_amplitude_gate_boundary(oracle):
    L = oracle.lib()
    assert L.bdo_amplitude_high_enough(9829, 9830) == 0
    assert L.bdo_amplitude_high_enough(9830, 9830) == 1


def test_window_max(oracle):
    # src/strategies/window_stats.rs:41-55
    assert oracle.window_max([0]) == 0
    assert oracle.window_max([-5, 3, 4]) == 5
    assert oracle.window_max([-32768]) == 32767      # i16::MIN is mapped to -32767
    assert oracle.window_max([-32768, 32767, 1]) == 32767
    rng = np.random.default_rng(1)
    for n in (1, 2, 17, 512, 1024, 4097):
        a = rng.integers(-32768, 32768, n).astype(np.int16)
        want = int(np.minimum(np.abs(a.astype(np.int32)), 32767).max())
        assert oracle.window_max(a) == want


def test_converters(oracle):
    # src/record.rs:151-166
    u = np.array([0, 1, 16383, 16384, 32767, 32768, 49150, 49151, 65535], dtype=np.uint16)
    got = oracle.u16_data_to_i16(u)
    want = ((u.astype(np.int64) - 16383 + 32768) % 65536 - 32768).astype(np.int16)  # wrapping `as i16`
    np.testing.assert_array_equal(got, want)
    assert got[0] == -16383 and got[-1] == -16384  # 65535-16383 = 49152 wraps
    f = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 1e-6, np.nan, np.inf, -np.inf, 0.99999], dtype=np.float32)
    got = oracle.f32_data_to_i16(f)
    want = [0, 32767, -32767, 16383, -16383, 32767, -32768, 0, 0, 32767, -32768, 32766]
    np.testing.assert_array_equal(got, np.array(want, dtype=np.int16))


def test_harness_rules(oracle):
    # lib.rs:319-324 downmix: ((l + r) as f32 / 2.0) as i16 truncates toward zero
    inter = np.array([3, 4, -3, -4, 32767, 32767, -32768, -32768, 1, -2], dtype=np.int16)
    np.testing.assert_array_equal(oracle.stereo_to_mono(inter), np.array([3, -3, 32767, -32768, 0], dtype=np.int16))
    # lib.rs:270-276: pads with `remainder` zeros
    assert oracle.harness_padded_len(2048, 1024) == 2048
    assert oracle.harness_padded_len(2646000, 1024) == 2646000 + 1008
    assert oracle.harness_padded_len(2646000, 1024) // 1024 == 2584


def test_literal_and_full_modes_agree(oracle):
    """bdo_is_beat (literal early returns) and bdo_is_beat_full (evaluates
    everything) must report the same beats."""
    rng = np.random.default_rng(7)
    n = 1024 * 120
    x = (rng.integers(-600, 600, n)).astype(np.int16)
    for start in range(5000, n - 4096, 23000):
        t = np.arange(4096)
        x[start:start + 4096] += (28000 * np.exp(-t / 1300.0) * np.sin(2 * np.pi * 60 * t / 44100)).astype(np.int16)
    for kind in (oracle.KIND_LPF, oracle.KIND_SPECTRUM):
        for L in (1024, 512, 700):
            _, _, b_full = oracle.run_stream(kind, 44100, x, L, full=True)
            _, _, b_lit = oracle.run_stream(kind, 44100, x, L, full=False)
            assert b_full == b_lit
            assert len(b_full) > 0
