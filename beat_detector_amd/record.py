"""Caller conveniences of the reference's `record` module (src/record.rs:168-198:
`audio_input_device_list`, `print_audio_input_device_configs`, `get_backends`) for the
sample sources this package can listen to.

The reference enumerates cpal capture devices.  Audio capture is not part of this
library: `start_listening` takes any object that delivers callbacks of samples
(`sampling_rate`, `sample_format`, `max_callback_len`, `read(capacity)`).  The
devices listed here are the ones that need no audio stack: 16-bit PCM WAV files,
raw PCM streams (a pipe from `arecord`, `ffmpeg`, ...) and the synthetic click
track of the benchmarks.  A cpal / PortAudio adapter is one more class with the
same four members."""
from __future__ import annotations

import glob
import os
import sys
import time
from typing import BinaryIO, Dict, Iterable, Optional

import numpy as np

from . import api

# what cpal reports as "default input config" for a device
_FMT_NAME = {api.FMT_I16: "I16", api.FMT_U16: "U16", api.FMT_F32: "F32"}


class InputDevice:
    """What `start_listening` needs from a device (record.rs:39-62 reads the same three
    things from cpal: sample rate, sample format, a stream of callbacks)."""

    name = "input device"
    sampling_rate = 44100
    sample_format = api.FMT_I16
    channels = 1
    max_callback_len = 4096
    callback_len = 512          # samples per delivered callback (cpal decides this per platform)

    def default_input_config(self) -> Dict[str, object]:
        return {"channels": self.channels, "sample_rate": self.sampling_rate,
                "buffer_size": self.callback_len, "sample_format": _FMT_NAME[self.sample_format]}

    def read(self, capacity: int) -> np.ndarray:    # empty array = end of stream
        raise NotImplementedError


class ArrayDevice(InputDevice):
    """Mono i16 samples held in memory, delivered `callback_len` at a time; `realtime`
    paces the callbacks like a capture device would."""

    def __init__(self, name: str, samples, sampling_rate: int = 44100, callback_len: int = 512, realtime: bool = False):
        self.name, self.sampling_rate, self.callback_len = name, int(sampling_rate), int(callback_len)
        self._samples = np.ascontiguousarray(samples, dtype=np.int16)
        self._pos, self._realtime, self._t0 = 0, realtime, None

    def read(self, capacity: int) -> np.ndarray:
        n = min(self.callback_len, int(capacity), self._samples.size - self._pos)
        if n <= 0:
            return np.zeros(0, dtype=np.int16)
        if self._realtime:
            self._t0 = self._t0 or time.monotonic()
            due = self._t0 + (self._pos + n) / self.sampling_rate
            time.sleep(max(0.0, due - time.monotonic()))
        out = self._samples[self._pos:self._pos + n]
        self._pos += n
        return out


class WavFileDevice(ArrayDevice):
    """A 16-bit PCM WAV file, down-mixed like the reference's harness (src/lib.rs:310-337)."""

    def __init__(self, path: str, callback_len: int = 512, realtime: bool = False):
        from .harness import read_wav_to_mono
        samples, rate = read_wav_to_mono(path)
        super().__init__(os.path.basename(path), samples, rate, callback_len, realtime)
        self.path = path


class SyntheticDevice(ArrayDevice):
    """The benchmarks' integer click track (bd_synth_fill_host), `seconds` long."""

    def __init__(self, seconds: float = 10.0, seed: int = 1, sampling_rate: int = 44100, callback_len: int = 512,
                 realtime: bool = False):
        n = int(seconds * sampling_rate)
        samples = api.synth_fill_host(1, n, first_stream=0, seed=seed, sampling_rate=sampling_rate)[0]
        super().__init__(f"synthetic click track (seed {seed}, {seconds:g} s)", samples, sampling_rate, callback_len, realtime)


class RawPcmDevice(InputDevice):
    """Mono PCM from a binary stream (stdin, a pipe, a socket file): i16 / u16 / f32, native byte order."""

    def __init__(self, stream: BinaryIO, sampling_rate: int = 44100, sample_format: int = api.FMT_I16,
                 callback_len: int = 512, name: str = "raw PCM stream"):
        self.name, self.sampling_rate, self.sample_format, self.callback_len = name, int(sampling_rate), sample_format, callback_len
        self._stream, self._dtype = stream, np.dtype(api._FMT_DTYPE[sample_format])

    def read(self, capacity: int) -> np.ndarray:
        n = min(self.callback_len, int(capacity))
        data = self._stream.read(n * self._dtype.itemsize)
        return np.frombuffer(data[:len(data) - len(data) % self._dtype.itemsize], dtype=self._dtype)


def audio_input_device_list(search: Optional[Iterable[str]] = None, callback_len: int = 512,
                            realtime: bool = False) -> Dict[str, InputDevice]:
    """record::audio_input_device_list (record.rs:168-177): name -> device, ordered by name
    like the reference's BTreeMap.  Lists the WAV files under `search` (directories or glob
    patterns; default: the BEATDET_INPUTS environment variable, ':'-separated) and the
    synthetic click track, so that the list is never empty."""
    if search is None:
        search = [p for p in os.environ.get("BEATDET_INPUTS", "").split(":") if p]
    devices: Dict[str, InputDevice] = {}
    for i, entry in enumerate(search):
        paths = sorted(glob.glob(os.path.join(entry, "*.wav"))) if os.path.isdir(entry) else sorted(glob.glob(entry))
        for path in paths:
            try:
                dev = WavFileDevice(path, callback_len, realtime)
            except Exception as exc:        # unreadable file: listed by the reference as "Unknown device #i"
                print(f"skipping {path}: {exc}", file=sys.stderr)
                continue
            devices[dev.name if dev.name not in devices else path] = dev
    synth = SyntheticDevice(callback_len=callback_len, realtime=realtime)
    devices[synth.name] = synth
    return dict(sorted(devices.items()))


def print_audio_input_device_configs(devices: Optional[Dict[str, InputDevice]] = None, file=None) -> None:
    """record::print_audio_input_device_configs (record.rs:179-191): one block per device on stderr."""
    file = file or sys.stderr
    for name, dev in (devices if devices is not None else audio_input_device_list()).items():
        print("--------", file=file)
        print(f"[{name}] default config:", file=file)
        cfg = dev.default_input_config()
        print("SupportedStreamConfig {", file=file)
        for key, value in cfg.items():
            print(f"    {key}: {value},", file=file)
        print("}", file=file)


def get_backends() -> Dict[str, object]:
    """record::get_backends (record.rs:193-198) names cpal's hosts; here: the kinds of input
    this module can open, and the GPUs the analysis can run on (empty without a GPU)."""
    backends: Dict[str, object] = {"WavFile": WavFileDevice, "RawPcm": RawPcmDevice, "Synthetic": SyntheticDevice}
    backends["Hip"] = [api.device_name(i)[0] for i in range(api.device_count())]
    return backends


def gpu_device_list() -> Dict[str, int]:
    """name -> index of every visible GPU, ordered by name."""
    return dict(sorted((f"[{i}] {api.device_name(i)[0]}", i) for i in range(api.device_count())))
