"""CPU tests of the definition of the extension onset analysis (oracle/bd_oracle_ext.c).

The reference computes none of this, so there is no reference vector to pin it
to: the f32 spectrum is checked against numpy's float64 FFT and against the
definition's own float64 evaluation, the rest against hand-computed cases.
"""
import numpy as np
import pytest

from tests.util import click_track


@pytest.mark.parametrize("n", [1024, 2048])
@pytest.mark.parametrize("window", [0, 1])
def test_spectrum_matches_float64_fft(oracle_ext, n, window):
    rng = np.random.default_rng(n + window)
    for amp in (30000, 300, 3):
        x = rng.integers(-amp, amp + 1, n).astype(np.int16)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n) if window else np.ones(n)
        want = np.abs(np.fft.rfft(x.astype(np.float64) * w))
        got = oracle_ext.rfft_mag(x, bool(window)).astype(np.float64)
        scale = np.sqrt(np.sum((x * w) ** 2))  # rounding noise is relative to the frame's energy
        assert np.max(np.abs(got - want)) <= 2e-6 * scale
        own = oracle_ext.rfft_mag_f64(x, bool(window))
        assert np.max(np.abs(own - want)) <= 1e-9 * scale


def test_pure_tone_lands_in_its_bin(oracle_ext):
    n, k = 2048, 37
    x = np.round(20000 * np.cos(2 * np.pi * k * np.arange(n) / n)).astype(np.int16)
    mag = oracle_ext.rfft_mag(x, True)
    # Hann: half of the amplitude * n / 2 in bin k, a quarter in k +- 1, nothing elsewhere
    assert abs(mag[k] - 20000 * n / 4) < 1e-3 * mag[k]
    assert abs(mag[k - 1] - 20000 * n / 8) < 1e-3 * mag[k]
    assert np.all(np.delete(mag, [k - 1, k, k + 1]) < 1e-4 * mag[k])


def test_frames_and_time_stamps(oracle_ext):
    assert oracle_ext.frames(2047, 2048, 1024) == 0
    assert oracle_ext.frames(2048, 2048, 1024) == 1
    assert oracle_ext.frames(3071, 2048, 1024) == 1
    assert oracle_ext.frames(3072, 2048, 1024) == 2
    assert oracle_ext.frames(44100 * 300, 2048, 1024) == 12918
    assert oracle_ext.frame_ms(0, 2048, 1024, 44100) == 23      # 1024 * 1000 / 44100
    assert oracle_ext.frame_ms(43, 2048, 1024, 44100) == 1021   # (43 * 1024 + 1024) * 1000 / 44100


def test_silence_and_constant_have_no_flux(oracle_ext):
    z = np.zeros(8192, dtype=np.int16)
    flux, band = oracle_ext.onset_stft(z, 44100)
    assert flux.shape == (7,) and not flux.any() and not band.any()
    c = np.full(8192, 1000, dtype=np.int16)
    flux, band = oracle_ext.onset_stft(c, 44100)
    assert flux[0] > 0 and np.all(flux[1:] == 0.0)           # the spectrum never changes after frame 0
    assert np.all(band == band[0]) and band[0] > 0


def test_band_energy_is_the_low_bins(oracle_ext):
    x = click_track(2048 * 4, 11)
    p = oracle_ext.default_params()
    _, band = oracle_ext.onset_stft(x, 44100, p)
    # bins with k * 44100 / 2048 <= 90 Hz: 0 .. 4
    mag = oracle_ext.rfft_mag(x[1024:3072], True).astype(np.float64)
    assert abs(band[1] - np.sum(mag[:5] ** 2)) <= 1e-5 * band[1]
    p48 = oracle_ext.default_params(fft_len=1024)
    _, band = oracle_ext.onset_stft(x, 48000, p48)
    mag = oracle_ext.rfft_mag(x[512:1536], True).astype(np.float64)
    assert abs(band[1] - np.sum(mag[:2] ** 2)) <= 1e-5 * band[1]   # 46.9 Hz per bin: bins 0, 1


def test_peak_picking_known_answer(oracle_ext):
    p = oracle_ext.default_params(mean_frames=2, max_frames=1, delta=0.5, lambda_=1.0, min_gap_ms=0)
    flux = np.array([0, 0, 5, 0, 0, 0, 3, 3, 0, 0, 0.4, 0], dtype=np.float32)
    cand, frame, ms = oracle_ext.onset_pick(flux, 44100, p)
    # j=2: mean(0,0,5,0,0)=1 -> thr 1.5 < 5, local max.  j=6: mean(0,0,3,3,0)=1.2, tie with j=7: earliest wins.
    # j=10: mean(0,0,.4,0)=.1 -> thr .6 > .4: rejected by the threshold.
    assert cand.tolist() == [0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0]
    assert frame.tolist() == [2, 6]
    assert ms.tolist() == [oracle_ext.frame_ms(2, 2048, 1024, 44100), oracle_ext.frame_ms(6, 2048, 1024, 44100)]
    # minimum gap: frames are 23.2 ms apart, frame 6 is 93 ms after frame 2
    p.min_gap_ms = 100
    _, frame, _ = oracle_ext.onset_pick(flux, 44100, p)
    assert frame.tolist() == [2]
    p.min_gap_ms = 93
    _, frame, _ = oracle_ext.onset_pick(flux, 44100, p)
    assert frame.tolist() == [2, 6]


def test_median_threshold_known_answer(oracle_ext):
    p = oracle_ext.default_params(mean_frames=2, max_frames=1, delta=0.5, lambda_=1.0, min_gap_ms=0, use_median=1)
    flux = np.array([0, 0, 5, 0, 0, 0, 3, 3, 0, 0, 0.4, 0], dtype=np.float32)
    cand, frame, _ = oracle_ext.onset_pick(flux, 44100, p)
    # medians of the 5-frame windows are 0 everywhere except around the 3, 3 pair: thresholds 0.5;
    # j = 10: 0.4 < 0.5 rejected; j = 6 wins the tie with j = 7
    assert frame.tolist() == [2, 6]
    # a spike does not lift a median threshold the way it lifts a mean threshold
    flux = np.array([1, 1, 1, 1, 50, 1, 1.6, 1, 1, 1, 1], dtype=np.float32)
    pm = oracle_ext.default_params(mean_frames=3, max_frames=1, delta=0.0, lambda_=1.5, min_gap_ms=0, use_median=0)
    pd = oracle_ext.default_params(mean_frames=3, max_frames=1, delta=0.0, lambda_=1.5, min_gap_ms=0, use_median=1)
    assert oracle_ext.onset_pick(flux, 44100, pm)[1].tolist() == [4]
    assert oracle_ext.onset_pick(flux, 44100, pd)[1].tolist() == [4, 6]


def test_clicks_are_found(oracle_ext):
    sr, period = 44100, 22050
    x = click_track(sr * 10, 5, sr=sr, period=period, noise=200)
    flux, _ = oracle_ext.onset_stft(x, sr)
    _, frame, ms = oracle_ext.onset_pick(flux, sr)
    assert 18 <= len(ms) <= 21
    gaps = np.diff(ms.astype(np.int64))
    # frame 0 is compared with silence, so its flux is its whole spectrum: skip the first gap
    assert np.all(np.abs(gaps[1:] - 500) <= 47)  # one hop is 23 ms


@pytest.mark.parametrize("n", [1024, 2048])
def test_quarter_hop_contains_the_half_hop_frames(oracle_ext, n):
    """hop = N / 4 adds a frame between every two of hop = N / 2: the band energy of its even frames
    is that of the half-hop analysis bit for bit (a frame's energy depends on the frame alone), the
    frame count and time stamps follow the hop, and other hops are refused."""
    x = click_track(44100 * 2 + 17, 5)
    half = oracle_ext.default_params(fft_len=n)
    quarter = oracle_ext.default_params(fft_len=n, hop=n // 4)
    _, band_h = oracle_ext.onset_stft(x, 44100, half)
    flux_q, band_q = oracle_ext.onset_stft(x, 44100, quarter)
    assert len(band_q) == oracle_ext.frames(len(x), n, n // 4) == (len(x) - n) // (n // 4) + 1
    assert np.array_equal(band_q[::2][:len(band_h)].view(np.uint32), band_h.view(np.uint32))
    assert (flux_q >= 0).all() and flux_q.max() > 0
    _, frame, ms = oracle_ext.onset_pick(flux_q, 44100, quarter)
    assert all(m == ((int(f) * (n // 4) + n // 2) * 1000) // 44100 for f, m in zip(frame, ms))
    with pytest.raises(Exception):
        oracle_ext.onset_stft(x, 44100, oracle_ext.default_params(fft_len=n, hop=n // 8))


def test_power_flux_is_the_flux_of_the_squared_magnitudes(oracle_ext):
    """flux_power: sum_k max(0, |X_j[k]|^2 - |X_j-1[k]|^2) with no square root anywhere; band energy is the
    same quantity in both modes.  Checked against float64 evaluations of the same frames."""
    x = click_track(2048 * 6, 21)
    for n, hop in ((2048, 1024), (1024, 256)):
        pm = oracle_ext.default_params(fft_len=n, hop=hop)
        pp = oracle_ext.default_params(fft_len=n, hop=hop, flux_power=1)
        flux_m, band_m = oracle_ext.onset_stft(x, 44100, pm)
        flux_p, band_p = oracle_ext.onset_stft(x, 44100, pp)
        assert flux_p.shape == flux_m.shape
        np.testing.assert_allclose(band_p, band_m, rtol=1e-5)
        prev = np.zeros(n // 2 + 1)
        for j in range(len(flux_p)):
            cur = oracle_ext.rfft_mag_f64(x[j * hop:j * hop + n], True) ** 2
            want = np.sum(np.maximum(cur - prev, 0.0))
            assert abs(flux_p[j] - want) <= 2e-5 * max(np.sum(cur), 1.0), j
            prev = cur
        # the onsets of a clean click track are found from either flux
        _, fm, _ = oracle_ext.onset_pick(flux_m, 44100, pm)
        _, fp, _ = oracle_ext.onset_pick(flux_p, 44100, pp)
        assert len(fp) > 0 and abs(len(fp) - len(fm)) <= 2
