"""Seeded differential sweep: random callback lengths, stream counts and lengths, row strides and
base offsets (rows that are only 2-byte aligned), both strategies and the documented parameter
variants, each compared with the CPU oracle.  Exercises every kernel the length routing can pick
(half-block with one, two and four callbacks per half-block, granule 64 / 32 / 16, split, window, generic, LPF aligned and
line staging) on inputs no
hand-written case lists."""
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SR = 44100


def _case(rng):
    kind_name = "Spectrum" if rng.random() < 0.6 else "LPF"
    pick = rng.random()
    if pick < 0.25:
        L = int(rng.choice([512, 1024]))
    elif pick < 0.45:
        L = 64 * int(rng.integers(1, 70))
    elif pick < 0.60:
        L = 16 * int(rng.integers(1, 40))
    elif pick < 0.75:
        L = int(rng.choice([441, 480, 882, 960, 1323, 160, 320, 100, 1000, 128, 256, 529, 200]))
    else:
        L = int(rng.integers(1, 5000))
    n_streams = int(rng.integers(1, 6))
    n = int(L * rng.integers(1, 12) + rng.integers(0, L)) if rng.random() < 0.3 else int(rng.integers(L, L + 70000))
    n = min(n, 140000)
    if n < L:
        n = L
    pad = int(rng.integers(0, 10))
    off = int(rng.integers(0, 8))
    if rng.random() < 0.3:          # rows on 16-byte boundaries: the kernels with aligned loads
        off, pad = 0, (-n) % 8
    sr = int(rng.choice([SR, SR, SR, 48000, 22050, 96000]))
    return kind_name, L, n_streams, n, pad, off, sr


@pytest.mark.parametrize("seed", range(int(os.environ.get("BD_FUZZ_SEEDS", "24"))))   # more seeds for a soak run
def test_random_shapes_against_the_oracle(bd, oracle, seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    variants = [dict(), dict(spectrum_include_dc=0), dict(spectrum_dc_packs_nyquist=1), dict(spectrum_divisor=2),
                dict(spectrum_max_hz=50.0)]
    for _ in range(12):
        kind_name, L, n_streams, n, pad, off, sr = _case(rng)
        kw = variants[int(rng.integers(0, len(variants)))] if kind_name == "Spectrum" else {}
        if sr == 96000 and "spectrum_include_dc" in kw:
            kw = {}   # bin 1 lies at 93.75 Hz: without the DC bin nothing is kept and the call is an error
        x = bd.synth_fill_host(n_streams, n, first_stream=int(rng.integers(0, 1000)), seed=int(rng.integers(1, 1 << 30)),
                               sampling_rate=sr)
        stride = n + pad
        flat = torch.full((off + n_streams * stride,), 12345, dtype=torch.int16, device="cuda")
        view = flat[off:].view(n_streams, stride)
        view[:, :n] = torch.from_numpy(x).cuda()
        kind = bd.StrategyKind[kind_name]
        okind = oracle.KIND_SPECTRUM if kind_name == "Spectrum" else oracle.KIND_LPF
        res = bd.analyze_batch(kind, sr, view, L, n_samples=n, params=bd.Params.default(**kw))
        torch.cuda.synchronize()
        p = oracle.default_params(**kw)
        ref_v, ref_c, ref_b = util.oracle_batch(oracle, okind, sr, x, L, p)
        got_v, got_c = res.values.cpu().numpy(), res.flags.cpu().numpy()
        ctx = f"{kind_name} L={L} streams={n_streams} n={n} pad={pad} off={off} sr={sr} {kw} kernel={bd.last_kernel_name()}"
        assert got_v.shape == ref_v.shape, ctx
        if kind_name == "LPF":
            np.testing.assert_array_equal(got_v, ref_v, err_msg=ctx)
            np.testing.assert_array_equal(got_c, ref_c, err_msg=ctx)
        else:
            nb, _, _ = oracle.bin_range(sr, p)
            div = {0: nb, 1: 1024, 2: 0}[p.spectrum_divisor]
            assert util.values_close(got_v, ref_v, np.abs(x.astype(np.int32)).max(), div).all(), ctx
            adjacent = util.threshold_adjacent(ref_v, p.spectrum_threshold)
            assert not ((got_c != ref_c) & ~adjacent).any(), ctx
        if (got_c == ref_c).all():
            for s in range(n_streams):
                assert res.beats(s) == ref_b[s], ctx
