"""The caller conveniences of the reference's record module and examples
(src/record.rs:168-198, examples/*.rs): device listing, config printing, the
console selection dialogue -- host logic, no GPU needed -- and, on a GPU, the
example's whole loop against the oracle."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def _write_wav(path, samples, rate=44100, channels=1):
    from scipy.io import wavfile
    data = np.asarray(samples, dtype=np.int16)
    wavfile.write(path, rate, data.reshape(-1, channels) if channels > 1 else data)


def test_device_list_is_ordered_by_name_and_never_empty(bd, tmp_path):
    from beat_detector_amd import record
    _write_wav(tmp_path / "b.wav", np.arange(100))
    _write_wav(tmp_path / "a.wav", np.arange(200), rate=48000)
    (tmp_path / "broken.wav").write_bytes(b"not a wav file")
    devs = record.audio_input_device_list([str(tmp_path)])
    assert list(devs) == sorted(devs) and len(devs) == 3            # BTreeMap order; the broken file is skipped
    assert list(devs)[:2] == ["a.wav", "b.wav"]
    assert devs["a.wav"].sampling_rate == 48000 and devs["a.wav"].sample_format == bd.FMT_I16
    only_synth = record.audio_input_device_list([])
    assert len(only_synth) == 1 and "synthetic" in next(iter(only_synth))


def test_wav_device_delivers_callbacks_and_downmixes_like_the_reference(bd, oracle, tmp_path):
    from beat_detector_amd import record
    rng = np.random.default_rng(3)
    stereo = rng.integers(-32768, 32768, 2 * 1000).astype(np.int16)
    _write_wav(tmp_path / "s.wav", stereo, channels=2)
    dev = record.WavFileDevice(str(tmp_path / "s.wav"), callback_len=300)
    got = []
    while True:
        block = dev.read(4096)
        if block.size == 0:
            break
        got.append(block)
    assert [len(b) for b in got] == [300, 300, 300, 100]
    assert np.array_equal(np.concatenate(got), oracle.stereo_to_mono(stereo))    # lib.rs:319-324
    assert dev.read(4096).size == 0                                              # stays at the end


def test_raw_pcm_device_reads_whole_samples(bd):
    from beat_detector_amd import record
    x = np.arange(-700, 700, dtype=np.int16)
    dev = record.RawPcmDevice(io.BytesIO(x.tobytes() + b"\x01"), callback_len=512)   # one stray byte at the end
    blocks = [dev.read(4096), dev.read(4096), dev.read(4096), dev.read(4096)]
    assert [b.size for b in blocks] == [512, 512, 376, 0]
    assert np.array_equal(np.concatenate(blocks), x)
    f = np.linspace(-1, 1, 100, dtype=np.float32)
    dev = record.RawPcmDevice(io.BytesIO(f.tobytes()), sample_format=bd.FMT_F32, callback_len=64)
    assert np.array_equal(np.concatenate([dev.read(64), dev.read(64)]), f)


def test_print_configs_and_backends(bd, tmp_path):
    from beat_detector_amd import record
    _write_wav(tmp_path / "x.wav", np.zeros(10))
    out = io.StringIO()
    record.print_audio_input_device_configs(record.audio_input_device_list([str(tmp_path)]), file=out)
    text = out.getvalue()
    assert text.count("--------") == 2 and "[x.wav] default config:" in text
    assert "sample_rate: 44100" in text and "sample_format: I16" in text and "channels: 1" in text
    backends = record.get_backends()
    assert {"WavFile", "RawPcm", "Synthetic", "Hip"} <= set(backends)
    assert len(backends["Hip"]) == bd.device_count()


def test_device_name_rejects_indices_outside_the_device_list(bd):
    with pytest.raises(bd.BeatDetectorError):
        bd.device_name(bd.device_count())
    with pytest.raises(bd.BeatDetectorError):
        bd.device_name(-1)


def test_console_dialogue_follows_the_reference(bd):
    import _common
    devs = {"a": "dev a", "b": "dev b", "c": "dev c"}
    out = io.StringIO()
    assert _common.select_input_device(devs, io.StringIO("1\n"), out) == "dev b"
    assert out.getvalue() == ("Available audio devices:\n  [0] a\n  [1] b\n  [2] c\n"
                              "Select audio device: input device number and enter:\n")
    out = io.StringIO()
    assert _common.select_strategy(io.StringIO("0\n"), out) == bd.StrategyKind.LPF
    assert _common.select_strategy(io.StringIO(" 1 \n"), io.StringIO()) == bd.StrategyKind.Spectrum
    lines = out.getvalue().splitlines()
    assert lines[0] == "Available beat detection strategies:" and lines[-1] == "Select strategy: input id and enter:"
    assert lines[1].startswith("  [0] Simple Lowpass Filter - ") and lines[2].startswith("  [1] Simple Spectrum Analysis - ")
    for bad, message in (("7\n", "Invalid strategy!"), ("x\n", "Input must be a valid number!"),
                         ("300\n", "Input must be a valid number!"), ("", "Input must be a valid number!")):
        with pytest.raises(SystemExit) as err:
            _common.select_strategy(io.StringIO(bad), io.StringIO())
        assert str(err.value) == message


@pytest.mark.gpu
@pytest.mark.parametrize("choice,kind_name", [("0", "LPF"), ("1", "Spectrum")])
def test_example_prints_the_beats_the_oracle_finds(bd, oracle, tmp_path, choice, kind_name):
    import audio_input_beat_detection as example
    from beat_detector_amd import record
    assert record.SyntheticDevice(seconds=1, seed=5).read(4096).size == 512
    x = bd.synth_fill_host(1, 44100 * 8, first_stream=0, seed=5)[0]
    _write_wav(tmp_path / "clicks.wav", x)
    os.environ["BEATDET_INPUTS"] = str(tmp_path)
    try:
        out = io.StringIO()
        example.main(["--fast"], io.StringIO(f"0\n{choice}\n"), out)   # device 0 = clicks.wav, then the strategy
    finally:
        del os.environ["BEATDET_INPUTS"]
    okind = oracle.KIND_LPF if kind_name == "LPF" else oracle.KIND_SPECTRUM
    _, _, beats = oracle.run_stream(okind, 44100, x, 512, full=True)
    want = [f"Found beat at BeatInfo {{ relative_ms: {ms} }}ms" for _, ms in beats]
    got = [line for line in out.getvalue().splitlines() if line.startswith("Found beat")]
    assert got == want and len(want) >= 3
    assert "[0] clicks.wav" in out.getvalue()
    assert bd.device_name(0)[0] and bd.device_name(0)[1] > 2 ** 30
