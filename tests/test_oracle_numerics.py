"""The oracle's restatements of the two third-party routines against
independent statements of the same published algorithms (numpy float32 /
float64) and against closed forms.  These are the PARITY-UNPINNED parts: the
tests pin the oracle to the documented hypotheses, not to the crates."""
import numpy as np
import pytest

SR = 44100


def _lpf_numpy(x, sr_u16, fc):
    """y0 = alpha*x0; y[i] = y[i-1] + alpha*(x[i]-y[i-1]), f32, truncated to i16 each step."""
    f = np.float32
    rc = f(1.0) / (f(fc) * f(2.0) * f(np.pi))
    dt = f(1.0) / f(sr_u16)
    alpha = dt / (rc + dt)
    y = np.zeros(len(x), dtype=np.int16)
    prev = f(0.0)
    for i, xi in enumerate(x):
        if i == 0:
            s = alpha * f(xi)
        else:
            d = f(xi) - prev
            m = f(alpha * d)
            s = f(prev + m)
        t = int(np.trunc(s))
        t = max(-32768, min(32767, t))
        y[i] = t
        prev = f(t)
    return y, alpha


def test_lpf_against_numpy_float32(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 2, 317, 1024):
        x = rng.integers(-32768, 32768, n).astype(np.int16)
        want, alpha = _lpf_numpy(x, SR, 120)
        np.testing.assert_array_equal(oracle.apply_lpf(x, SR, 120), want)
    assert abs(float(alpha) - 0.016809706) < 1e-8


def test_lpf_step_and_impulse(oracle):
    step = np.full(1024, 20000, dtype=np.int16)
    y = oracle.apply_lpf(step, SR, 120)
    assert y[0] == int(0.016809706 * 20000)             # 336
    assert (np.diff(y.astype(np.int32)) >= 0).all()     # monotone rise
    assert 19000 < y[-1] < 20000                        # truncation keeps it below the input
    imp = np.zeros(64, dtype=np.int16); imp[0] = 32767
    y = oracle.apply_lpf(imp, SR, 120)
    assert y[0] == 550 and (np.diff(y.astype(np.int32)) <= 0).all() and y[-1] >= 0


def test_lpf_sampling_rate_wraps_as_u16(oracle):
    x = (20000 * np.sin(2 * np.pi * 60 * np.arange(1024) / 88200)).astype(np.int16)
    want, _ = _lpf_numpy(x, 88200 & 0xFFFF, 120)        # 22664, src/strategies/lpf.rs:69
    np.testing.assert_array_equal(oracle.apply_lpf(x, 88200, 120), want)


def test_spectrum_bins_kept(oracle):
    # bins at k * fs/1024 <= 90 Hz: 0, 43.07, 86.13 at 44.1 kHz
    assert oracle.bin_range(44100) == (3, 0, 2)
    assert oracle.bin_range(48000) == (2, 0, 1)
    assert oracle.bin_range(96000) == (1, 0, 0)
    assert oracle.bin_range(8000) == (12, 0, 11)
    assert oracle.bin_range(44100, oracle.default_params(spectrum_include_dc=0)) == (2, 1, 2)
    # 46080 Hz puts bin 2 exactly on 90 Hz: inclusive keeps it, exclusive drops it
    assert oracle.bin_range(46080) == (3, 0, 2)
    assert oracle.bin_range(46080, oracle.default_params(spectrum_inclusive=0)) == (2, 0, 1)


def test_spectrum_pure_tones_have_analytic_magnitudes(oracle):
    n = np.arange(1024)
    for k, amp in ((1, 20000.0), (2, 12345.0)):
        ring = (amp * np.cos(2 * np.pi * k * n / 1024)).astype(np.float32)
        v, nb = oracle.spectrum_band_max(ring, SR)
        assert nb == 3
        assert v == pytest.approx(amp * 1024 / 2 / 3, rel=1e-5)     # A*N/2, divided by the 3 kept bins
    ring = np.full(1024, 1000.0, dtype=np.float32)
    v, _ = oracle.spectrum_band_max(ring, SR)
    assert v == pytest.approx(1000.0 * 1024 / 3, rel=1e-6)           # DC bin
    v, _ = oracle.spectrum_band_max(ring, SR, oracle.default_params(spectrum_divisor=oracle.DIV_FFT_LEN))
    assert v == pytest.approx(1000.0, rel=1e-6)
    v, _ = oracle.spectrum_band_max(ring, SR, oracle.default_params(spectrum_divisor=oracle.DIV_NONE))
    assert v == pytest.approx(1000.0 * 1024, rel=1e-6)
    v, _ = oracle.spectrum_band_max(ring, SR, oracle.default_params(spectrum_include_dc=0))
    assert v < 1.0                                                   # nothing but DC in the signal


def test_spectrum_fft_against_float64_dft(oracle):
    rng = np.random.default_rng(2)
    for _ in range(20):
        ring = rng.integers(-32768, 32768, 1024).astype(np.float32)
        for over in ({}, {"spectrum_dc_packs_nyquist": 1}, {"spectrum_include_dc": 0}):
            p = oracle.default_params(**over)
            v32, _ = oracle.spectrum_band_max(ring, SR, p)
            v64 = oracle.spectrum_band_max_f64(ring, SR, p)
            assert v32 == pytest.approx(v64, rel=1e-5, abs=1.0)
        X = np.fft.rfft(ring.astype(np.float64))
        assert oracle.spectrum_band_max_f64(ring, SR) == pytest.approx(np.abs(X[:3]).max() / 3, rel=1e-12)


def test_divisor_fft_length_can_never_fire(oracle):
    """With i16 input |X_k| <= 1024 * 32768, so value / 1024 <= 32768 < 2.1e6:
    the documented reason for the default divisor."""
    ring = np.full(1024, -32768.0, dtype=np.float32)
    v, _ = oracle.spectrum_band_max(ring, SR, oracle.default_params(spectrum_divisor=oracle.DIV_FFT_LEN))
    assert v == 32768.0
    v, _ = oracle.spectrum_band_max(ring, SR)
    assert v > 2_100_000.0


def test_ring_is_chronological_for_partial_callbacks(oracle):
    """spectrum.rs:65-68: with 529-sample callbacks the ring holds the last 1024
    samples in arrival order; the amplitude gate looks at the callback only."""
    rng = np.random.default_rng(4)
    x = rng.integers(-20000, 20000, 529 * 7).astype(np.int16)
    d = oracle.OracleDetector(oracle.KIND_SPECTRUM, SR)
    for j in range(7):
        d.is_beat(x[529 * j:529 * (j + 1)], full=True)
        end = 529 * (j + 1)
        ring = np.zeros(1024, dtype=np.float32)
        take = min(end, 1024)
        ring[1024 - take:] = x[end - take:end]
        want, _ = oracle.spectrum_band_max(ring, SR)
        assert d.last_value == want
