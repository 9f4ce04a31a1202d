"""GPU tests of the carried-state biquad envelope (bd_biquad_envelope) against its
CPU definition: values and decisions bit-identical."""
import numpy as np
import pytest

from tests.util import click_track

pytestmark = pytest.mark.gpu
SR = 44100


def _rows(n_streams, n, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n_streams):
        k = i % 3
        if k == 0:
            rows.append(click_track(n, seed * 31 + i))
        elif k == 1:
            rows.append(rng.integers(-32768, 32768, n).astype(np.int16))
        else:
            t = np.arange(n)
            rows.append(np.round(25000 * np.sin(2 * np.pi * 70.0 * t / SR) * (np.sin(2 * np.pi * 1.5 * t / SR) > 0))
                        .astype(np.int16))
    return rows


def _to_dev(rows):
    import torch
    n = max(len(r) for r in rows)
    host = np.zeros((len(rows), (n + 63) & ~63), dtype=np.int16)
    for i, r in enumerate(rows):
        host[i, :len(r)] = r
    return torch.from_numpy(host).cuda(), host


@pytest.mark.parametrize("window_len", [1024, 64, 256, 4096])
@pytest.mark.parametrize("n", [4096 * 5 + 1024 * 2 + 77, 4096 * 2, 4096 - 64, 64])
def test_values_and_flags_bit_exact(bd, oracle_ext, window_len, n):
    rows = _rows(7, n, seed=window_len + n)
    dev, host = _to_dev(rows)
    p = bd.BiquadParams.default(window_len=window_len)
    op = oracle_ext.biquad_params(window_len=window_len)
    values, flags = bd.biquad_envelope(SR, dev, n_samples=n, params=p)
    values, flags = values.cpu().numpy(), flags.cpu().numpy()
    assert values.shape == (7, n // window_len)
    for s in range(7):
        want_v, want_f = oracle_ext.biquad_envelope(host[s, :n], SR, op)
        assert np.array_equal(values[s].view(np.uint32), want_v.view(np.uint32)), (s, np.abs(values[s] - want_v).max())
        assert np.array_equal(flags[s], want_f), s


def test_other_rates_and_corners(bd, oracle_ext):
    n = 4096 * 4 + 1024
    dev, host = _to_dev(_rows(5, n, seed=3))
    for sr, fc, q, ratio, amp in ((48000, 120, 0.70710678, 0.77, 9830), (22050, 300, 1.5, 0.5, 100), (96000, 60, 0.5, 0.9, 0)):
        p = bd.BiquadParams.default(cutoff_hz=fc, q_factor=q, ratio=ratio, min_amplitude=amp)
        op = oracle_ext.biquad_params(cutoff_hz=fc, q_factor=q, ratio=ratio, min_amplitude=amp)
        values, flags = bd.biquad_envelope(sr, dev, n_samples=n, params=p)
        for s in range(5):
            want_v, want_f = oracle_ext.biquad_envelope(host[s, :n], sr, op)
            assert np.array_equal(values[s].cpu().numpy().view(np.uint32), want_v.view(np.uint32))
            assert np.array_equal(flags[s].cpu().numpy(), want_f)
    # an explicit design overrides the parameters' corner
    d = bd.biquad_design(SR, 500, 2.0)
    values, _ = bd.biquad_envelope(SR, dev, n_samples=n, design=d)
    od = oracle_ext.biquad_design(SR, 500, 2.0)
    want_v, _ = oracle_ext.biquad_envelope(host[0, :n], SR, design=od)
    assert np.array_equal(values[0].cpu().numpy().view(np.uint32), want_v.view(np.uint32))


def test_short_inputs_and_errors(bd):
    import torch
    dev = torch.zeros((3, 1024), dtype=torch.int16, device="cuda")
    values, flags = bd.biquad_envelope(SR, dev, n_samples=1023)
    assert values.shape == (3, 0)
    with pytest.raises(bd.BeatDetectorError):
        bd.biquad_envelope(SR, dev, params=bd.BiquadParams.default(window_len=1000))
    with pytest.raises(bd.BeatDetectorError):
        bd.biquad_envelope(SR, dev, params=bd.BiquadParams.default(cutoff_hz=30000))


def test_beats_from_the_envelope(bd, oracle_ext, oracle):
    """Envelope decisions + the reference's refractory rule (src/lib.rs:115-127) = beat list."""
    import torch
    n = SR * 20
    dev = torch.empty((300, (n + 63) & ~63), dtype=torch.int16, device="cuda")
    bd.synth_fill(dev, n_samples=n, seed=5, sampling_rate=SR)
    values, flags, count, idx, ms = bd.biquad_beats(SR, dev, n_samples=n)
    host = dev.cpu().numpy()
    count = count.cpu().numpy()
    for s in (0, 7, 150, 299):
        want_v, want_f = oracle_ext.biquad_envelope(host[s, :n], SR)
        assert np.array_equal(values[s].cpu().numpy().view(np.uint32), want_v.view(np.uint32))
        assert np.array_equal(flags[s].cpu().numpy(), want_f)
        # replay of the refractory rule on the definition's decisions
        st = oracle.State()
        oracle.lib().bdo_state_new(st, SR)
        beats = []
        for k, f in enumerate(want_f):
            oracle.lib().bdo_state_update_time(st, 1024)
            if oracle.lib().bdo_last_beat_beyond_threshold(st, 400) and f:
                oracle.lib().bdo_state_mark_beat(st)
                beats.append((k, st.beat_time_ms))
        assert count[s] == len(beats) > 0
        got = list(zip(idx[s, :count[s]].cpu().tolist(), ms[s, :count[s]].cpu().tolist()))
        assert got == beats


def test_rows_that_are_only_2_byte_aligned(bd, oracle_ext):
    """Odd strides and an odd base offset: same results through the unaligned loads."""
    import torch
    n = 4096 * 3 + 1024 + 9
    rows = _rows(5, n, seed=21)
    stride = n + 3                                   # odd: rows start on odd sample offsets
    flat = torch.zeros(1 + 5 * stride, dtype=torch.int16, device="cuda")
    view = flat[1:].view(5, stride)                  # base 2 bytes off a 16-byte boundary
    for i, r in enumerate(rows):
        view[i, :n] = torch.from_numpy(r).cuda()
    assert view.data_ptr() % 16 == 2
    values, flags = bd.biquad_envelope(SR, view, n_samples=n)
    for s in range(5):
        want_v, want_f = oracle_ext.biquad_envelope(rows[s], SR)
        assert np.array_equal(values[s].cpu().numpy().view(np.uint32), want_v.view(np.uint32)), s
        assert np.array_equal(flags[s].cpu().numpy(), want_f), s
