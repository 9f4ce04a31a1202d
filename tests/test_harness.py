"""The offline harness mirror (src/lib.rs:263-337 of the reference)."""
import numpy as np
import pytest


def test_downmix_and_padding_match_the_oracle(oracle):
    from beat_detector_amd import harness
    rng = np.random.default_rng(0)
    inter = rng.integers(-32768, 32768, 20000).astype(np.int16)
    inter[:8] = [32767, 32767, -32768, -32768, 1, -2, -1, 2]
    np.testing.assert_array_equal(harness.stereo_to_mono(inter), oracle.stereo_to_mono(inter))
    for n in (0, 5, 1024, 1025, 2646000):
        assert harness.pad_like_the_reference(np.zeros(n, np.int16), 1024).size == oracle.harness_padded_len(n, 1024)


def test_wav_reader(tmp_path):
    from scipy.io import wavfile
    from beat_detector_amd import harness
    rng = np.random.default_rng(1)
    stereo = rng.integers(-32768, 32768, (5000, 2)).astype(np.int16)
    wavfile.write(tmp_path / "s.wav", 44100, stereo)
    mono, rate = harness.read_wav_to_mono(str(tmp_path / "s.wav"))
    assert rate == 44100
    np.testing.assert_array_equal(mono, harness.stereo_to_mono(stereo.reshape(-1)))
    wavfile.write(tmp_path / "m.wav", 22050, stereo[:, 0].copy())
    mono, rate = harness.read_wav_to_mono(str(tmp_path / "m.wav"))
    assert rate == 22050 and (mono == stereo[:, 0]).all()


@pytest.mark.gpu
def test_harness_on_a_synthetic_clip_like_sample_1(bd, oracle):
    """The shape of the reference's test_sample_1_print_beats (lib.rs:198-219): an
    8 s 44.1 kHz clip, 1024-sample windows, both strategies.  (res/sample_1.mp3
    itself cannot be decoded here: no MP3 decoder.)"""
    from beat_detector_amd import harness
    x = bd.synth_fill_host(1, 44100 * 8 + 317, first_stream=1, seed=2021)[0]
    got = harness.apply_samples_to_all_strategies(1024, x, 44100)
    padded = harness.pad_like_the_reference(x, 1024)
    for kind, okind in ((bd.StrategyKind.LPF, oracle.KIND_LPF), (bd.StrategyKind.Spectrum, oracle.KIND_SPECTRUM)):
        _, _, beats = oracle.run_stream(okind, 44100, padded, 1024, full=False)
        assert [b.relative_ms() for b in got[kind]] == [ms for _, ms in beats]
        assert len(beats) > 0
