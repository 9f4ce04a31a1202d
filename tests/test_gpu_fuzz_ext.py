"""Seeded sweep of the extension analyses (onset STFT + peak picking, biquad envelope) over random
stream counts, lengths, row strides, base offsets (rows that are only 2-byte aligned) and parameters:
bit-identical to their CPU definitions."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _view(x, pad, off):
    import torch
    n_streams, n = x.shape
    stride = n + pad
    flat = torch.full((off + n_streams * stride + 8,), -7, dtype=torch.int16, device="cuda")
    view = flat[off:off + n_streams * stride].view(n_streams, stride)
    view[:, :n] = torch.from_numpy(x).cuda()
    return view


def _signal(rng, n_streams, n):
    rows = []
    for _ in range(n_streams):
        kind = int(rng.integers(0, 4))
        if kind == 0:
            rows.append(rng.integers(-32768, 32768, n).astype(np.int16))
        elif kind == 1:
            rows.append(rng.integers(-2, 3, n).astype(np.int16))
        elif kind == 2:
            rows.append(np.zeros(n, dtype=np.int16))
        else:
            t = np.arange(n)
            f = float(rng.uniform(30, 400))
            env = (np.sin(2 * np.pi * rng.uniform(0.5, 4) * t / 44100) > 0.3)
            rows.append(np.round(rng.uniform(2000, 32000) * np.sin(2 * np.pi * f * t / 44100) * env).astype(np.int16))
    return np.stack(rows)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BD_FUZZ_SEEDS", "8"))))
def test_onset_random_shapes(bd, oracle_ext, seed):
    rng = np.random.default_rng(5000 + seed)
    for _ in range(6):
        fft_len = int(rng.choice([1024, 2048]))
        hop = fft_len // int(rng.choice([2, 4]))
        n = int(rng.integers(fft_len - 3, fft_len * 40))
        n_streams = int(rng.integers(1, 6))
        sr = int(rng.choice([44100, 48000, 22050]))
        kw = dict(fft_len=fft_len, hop=hop, window=int(rng.integers(0, 2)), use_median=int(rng.integers(0, 2)),
                  flux_power=int(rng.integers(0, 2)),
                  mean_frames=int(rng.integers(0, 12)), max_frames=int(rng.integers(0, 5)),
                  min_gap_ms=int(rng.integers(0, 400)), band_max_hz=float(rng.choice([90.0, 200.0, 3000.0])))
        x = _signal(rng, n_streams, n)
        view = _view(x, int(rng.integers(0, 9)), int(rng.integers(0, 8)))
        res = bd.onset_batch(sr, view, n_samples=n, params=bd.OnsetParams.default(**kw), max_onsets=4096)
        op = oracle_ext.default_params(**kw)
        ctx = f"n={n} streams={n_streams} sr={sr} {kw}"
        for s in range(n_streams):
            flux, band = oracle_ext.onset_stft(x[s], sr, op)
            assert np.array_equal(res.flux[s].cpu().numpy().view(np.uint32), flux.view(np.uint32)), ctx
            assert np.array_equal(res.band_energy[s].cpu().numpy().view(np.uint32), band.view(np.uint32)), ctx
            _, frame, ms = oracle_ext.onset_pick(flux, sr, op)
            assert res.onsets(s) == list(zip(frame.tolist(), ms.tolist())), ctx


@pytest.mark.parametrize("seed", range(int(os.environ.get("BD_FUZZ_SEEDS", "8"))))
def test_biquad_random_shapes(bd, oracle_ext, seed):
    rng = np.random.default_rng(7000 + seed)
    for _ in range(6):
        window_len = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096]))
        n = int(rng.integers(1, 4096 * 12))
        n_streams = int(rng.integers(1, 6))
        sr = int(rng.choice([44100, 48000, 22050, 96000]))
        kw = dict(window_len=window_len, cutoff_hz=int(rng.choice([60, 120, 300, 1000])),
                  q_factor=float(rng.choice([0.5, 0.70710678, 2.0])), ratio=float(rng.choice([0.5, 0.77, 0.95])),
                  min_amplitude=int(rng.choice([0, 100, 9830])))
        x = _signal(rng, n_streams, n)
        view = _view(x, int(rng.integers(0, 9)), int(rng.integers(0, 8)))
        values, flags = bd.biquad_envelope(sr, view, n_samples=n, params=bd.BiquadParams.default(**kw))
        op = oracle_ext.biquad_params(**kw)
        ctx = f"n={n} streams={n_streams} sr={sr} {kw}"
        assert values.shape == (n_streams, n // window_len), ctx
        for s in range(n_streams):
            want_v, want_f = oracle_ext.biquad_envelope(x[s], sr, op)
            assert np.array_equal(values[s].cpu().numpy().view(np.uint32), want_v.view(np.uint32)), ctx
            assert np.array_equal(flags[s].cpu().numpy(), want_f), ctx
