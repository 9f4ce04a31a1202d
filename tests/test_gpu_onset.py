"""GPU tests of the extension onset analysis (bd_onset_* of the C ABI) against
its CPU definition.  The kernels follow the definition's f32 operation order, so
flux, band energy and onsets must be bit-identical."""
import numpy as np
import pytest

from tests.util import click_track

pytestmark = pytest.mark.gpu
SR = 44100


def _aligned_rows(rows, stride=None):
    import torch
    n = max(len(r) for r in rows)
    stride = stride or ((n + 63) & ~63)
    host = np.zeros((len(rows), stride), dtype=np.int16)
    for i, r in enumerate(rows):
        host[i, :len(r)] = r
    return torch.from_numpy(host).cuda(), host


def _streams(n_streams, n, seed=0):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n_streams):
        kind = i % 4
        if kind == 0:
            rows.append(click_track(n, seed * 100 + i))
        elif kind == 1:
            rows.append(rng.integers(-32768, 32768, n).astype(np.int16))
        elif kind == 2:
            rows.append(rng.integers(-3, 4, n).astype(np.int16))
        else:
            t = np.arange(n)
            rows.append(np.round(12000 * np.sin(2 * np.pi * 55.0 * t / SR) * (1 + 0.5 * np.sin(2 * np.pi * 2 * t / SR)))
                        .astype(np.int16))
    return rows


@pytest.mark.parametrize("fft_len", [1024, 2048])
@pytest.mark.parametrize("window", [1, 0])
@pytest.mark.parametrize("n", [2048 * 70 + 13, 2048 * 3, 2048])
@pytest.mark.parametrize("hop_div", [2, 4])
@pytest.mark.parametrize("power", [0, 1])
def test_flux_and_band_energy_bit_exact(bd, oracle_ext, fft_len, window, n, hop_div, power):
    rows = _streams(9, n, seed=fft_len + window)
    dev, host = _aligned_rows(rows)
    hop = fft_len // hop_div
    p = bd.OnsetParams.default(fft_len=fft_len, hop=hop, window=window, flux_power=power)
    op = oracle_ext.default_params(fft_len=fft_len, hop=hop, window=window, flux_power=power)
    flux, band = bd.onset_stft(SR, dev, n_samples=n, params=p)
    flux, band = flux.cpu().numpy(), band.cpu().numpy()
    nf = oracle_ext.frames(n, fft_len, hop)
    assert flux.shape == (9, nf)
    for s in range(9):
        want_flux, want_band = oracle_ext.onset_stft(host[s, :n], SR, op)
        assert np.array_equal(flux[s].view(np.uint32), want_flux.view(np.uint32)), (s, np.abs(flux[s] - want_flux).max())
        assert np.array_equal(band[s].view(np.uint32), want_band.view(np.uint32)), s


def test_short_and_empty_inputs(bd, oracle_ext):
    import torch
    dev = torch.zeros((3, 2048), dtype=torch.int16, device="cuda")
    flux, band = bd.onset_stft(SR, dev, n_samples=2047)
    assert flux.shape == (3, 0)
    res = bd.onset_batch(SR, dev, n_samples=100)
    assert res.onset_count.cpu().tolist() == [0, 0, 0]
    res = bd.onset_batch(SR, dev)       # one frame of silence
    assert res.flux.cpu().tolist() == [[0.0]] * 3 and res.onset_count.cpu().tolist() == [0, 0, 0]
    with pytest.raises(bd.BeatDetectorError):
        bd.onset_stft(SR, dev, params=bd.OnsetParams.default(fft_len=4096))
    with pytest.raises(bd.BeatDetectorError):
        bd.onset_stft(SR, dev, params=bd.OnsetParams.default(fft_len=2048, hop=256))


@pytest.mark.parametrize("fft_len", [1024, 2048])
def test_onsets_match_definition(bd, oracle_ext, fft_len):
    n = SR * 12 + 77
    rows = _streams(13, n, seed=7)
    dev, host = _aligned_rows(rows)
    for kw in (dict(), dict(hop=fft_len // 4), dict(hop=fft_len // 4, use_median=1, min_gap_ms=120),
               dict(mean_frames=3, max_frames=1, delta=1000.0, lambda_=1.2, min_gap_ms=250),
               dict(mean_frames=0, max_frames=0, delta=0.0, lambda_=0.5, min_gap_ms=0),
               dict(use_median=1), dict(use_median=1, mean_frames=32, max_frames=2, lambda_=2.0, min_gap_ms=50),
               dict(flux_power=1), dict(flux_power=1, hop=fft_len // 4, window=0)):
        p = bd.OnsetParams.default(fft_len=fft_len, **kw)
        op = oracle_ext.default_params(fft_len=fft_len, **kw)
        res = bd.onset_batch(SR, dev, n_samples=n, params=p)
        picked = bd.onset_pick(SR, res.flux, params=p)
        flux = res.flux.cpu().numpy()
        for s in range(13):
            cand, frame, ms = oracle_ext.onset_pick(flux[s], SR, op)
            assert res.onsets(s) == list(zip(frame.tolist(), ms.tolist())), (s, kw)
            assert picked.onsets(s) == res.onsets(s)
            assert np.array_equal(picked.candidates[s].cpu().numpy(), cand)


def test_max_onsets_truncates_but_counts_all(bd, oracle_ext):
    n = SR * 8
    dev, host = _aligned_rows([click_track(n, 3, period=11025)])
    res = bd.onset_batch(SR, dev, max_onsets=4)
    flux, _ = oracle_ext.onset_stft(host[0, :n], SR)
    _, frame, ms = oracle_ext.onset_pick(flux, SR)
    assert int(res.onset_count[0]) == len(frame) > 4
    assert res.onset_frame[0].cpu().tolist() == frame[:4].tolist()


def test_large_batch_is_consistent(bd, oracle_ext):
    """Many streams x long rows: every stream gets the result of the same stream alone
    (checked against the definition on a sample of rows)."""
    import torch
    n = SR * 30
    dev = torch.empty((512, (n + 63) & ~63), dtype=torch.int16, device="cuda")
    bd.synth_fill(dev, n_samples=n, seed=99, sampling_rate=SR)
    res = bd.onset_batch(SR, dev, n_samples=n)
    torch.cuda.synchronize()
    host = dev.cpu().numpy()
    for s in (0, 1, 255, 511):
        want_flux, want_band = oracle_ext.onset_stft(host[s, :n], SR)
        assert np.array_equal(res.flux[s].cpu().numpy().view(np.uint32), want_flux.view(np.uint32))
        assert np.array_equal(res.band_energy[s].cpu().numpy().view(np.uint32), want_band.view(np.uint32))
        _, frame, ms = oracle_ext.onset_pick(want_flux, SR)
        assert res.onsets(s) == list(zip(frame.tolist(), ms.tolist()))


def test_rows_that_are_only_2_byte_aligned(bd, oracle_ext):
    """Odd strides and an odd base offset: same results (dword loads at 2-byte-aligned addresses)."""
    import torch
    n = 2048 * 9 + 31
    rows = _streams(5, n, seed=4)
    stride = n + 1
    flat = torch.zeros(1 + 5 * stride, dtype=torch.int16, device="cuda")
    view = flat[1:].view(5, stride)
    for i, r in enumerate(rows):
        view[i, :n] = torch.from_numpy(r).cuda()
    assert view.data_ptr() % 4 == 2
    for fft_len in (1024, 2048):
        res = bd.onset_batch(SR, view, n_samples=n, params=bd.OnsetParams.default(fft_len=fft_len))
        op = oracle_ext.default_params(fft_len=fft_len)
        for s in range(5):
            flux, band = oracle_ext.onset_stft(rows[s], SR, op)
            assert np.array_equal(res.flux[s].cpu().numpy().view(np.uint32), flux.view(np.uint32)), s
            _, frame, ms = oracle_ext.onset_pick(flux, SR, op)
            assert res.onsets(s) == list(zip(frame.tolist(), ms.tolist()))
