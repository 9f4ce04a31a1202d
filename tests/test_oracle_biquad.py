"""CPU tests of the definition of the carried-state biquad envelope (extension
analysis, no reference counterpart) and of the library's host-side design."""
import ctypes as C

import numpy as np
import pytest

from tests.util import click_track


def test_design_matches_between_library_and_definition(bd, oracle_ext):
    for sr, fc, q in ((44100, 120, 0.70710678), (48000, 120, 0.70710678), (22050, 200, 1.3), (96000, 60, 0.5)):
        mine = bd.biquad_design(sr, fc, q)
        ref = oracle_ext.biquad_design(sr, fc, q)
        assert bytes(mine) == bytes(ref), (sr, fc, q)
    with pytest.raises(bd.BeatDetectorError):
        bd.biquad_design(44100, 30000)
    with pytest.raises(ValueError):
        oracle_ext.biquad_design(44100, 0)


def test_design_is_a_unit_gain_lowpass(oracle_ext):
    d = oracle_ext.biquad_design(44100)
    # DC gain (b0 + b1 + b2) / (1 - na1 - na2) = 1
    assert abs((d.b0 + d.b1 + d.b2) / (1.0 - d.na1 - d.na2) - 1.0) < 2e-3
    # chunk transition: q[0] = pw[0] = P, q[1] = pw[1] = P^2, q[3] = pw[2]
    assert list(d.q[0]) == list(d.pw[0]) and list(d.q[1]) == list(d.pw[1]) and list(d.q[3]) == list(d.pw[2])


@pytest.mark.parametrize("n", [4096 * 3, 4096 * 2 + 1024 * 3 + 5, 1024, 64 * 17])
def test_chunked_evaluation_follows_the_float64_filter(oracle_ext, n):
    x = click_track(n, n)
    p = oracle_ext.biquad_params(window_len=64)
    _, _, y = oracle_ext.biquad_envelope(x, 44100, p, want_y=True)
    y64 = oracle_ext.biquad_filter_f64(x, 44100)
    m = len(y)
    assert m == (n // 64) * 64
    # f32 direct form at a 120 Hz corner: coefficient rounding dominates; 3e-4 of the signal's peak
    assert np.max(np.abs(y - y64[:m])) <= 3e-4 * max(np.max(np.abs(y64)), 1.0)


def test_state_is_carried_across_windows(oracle_ext):
    # a step input: the second-order response keeps rising across the first windows,
    # which a filter restarted per window could not do
    x = np.full(8192, 10000, dtype=np.int16)
    p = oracle_ext.biquad_params(window_len=64, min_amplitude=0)
    v, f = oracle_ext.biquad_envelope(x, 44100, p)
    # rise over the first three 64-sample windows, 4.3 % Butterworth overshoot, then settles at the input level
    assert v[0] < 4000 < v[1] < 8000 < v[2] and 10400 < v[3:5].max() < 10500 and abs(v[-1] - 10000) < 5


def test_windows_gate_and_threshold(oracle_ext):
    sr = 44100
    t = np.arange(sr)
    quiet = (3000 * np.sin(2 * np.pi * 60 * t / sr)).astype(np.int16)       # below the amplitude gate
    bass = (20000 * np.sin(2 * np.pi * 60 * t / sr)).astype(np.int16)       # passes the filter
    treble = (20000 * np.sin(2 * np.pi * 4000 * t / sr)).astype(np.int16)   # removed by the filter
    for x, expect in ((quiet, 0), (bass, 1), (treble, 0)):
        v, f = oracle_ext.biquad_envelope(x, sr)
        assert f[5:].min() == f[5:].max() == expect
    # window lengths are powers of two from 64 to 4096
    for w in (32, 96, 8192):
        with pytest.raises(ValueError):
            oracle_ext.biquad_envelope(bass, sr, oracle_ext.biquad_params(window_len=w))
