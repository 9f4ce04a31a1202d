"""ctypes binding of libbeatdet_hip.so and the Python mirror of the reference's
per-window interface (src/lib.rs:51-185 of the reference).

torch is only plumbing here: device memory (tensors whose ``data_ptr()`` is
handed to the library) and the current HIP stream.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import enum
import os
from typing import List, Optional, Sequence

import numpy as np

try:  # torch must be loaded first so that both share one HIP runtime
    import torch
except Exception:  # pragma: no cover - torch is present in the target image
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libbeatdet_hip.so"

BD_OK = 0
BD_ERR_INVALID_ARG = -1
BD_ERR_NO_DEVICE = -2
BD_ERR_HIP = -3
BD_ERR_UNSUPPORTED = -4
BD_ERR_EMPTY_WINDOW = -5
BD_ERR_NO_BINS = -6
BD_ERR_STOPPED = -7

BD_DIV_BINS, BD_DIV_FFT_LEN, BD_DIV_NONE = 0, 1, 2


class BeatDetectorError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"beat_detector_amd: status {status}: {message}")
        self.status = status


class Params(C.Structure):
    """bd_params of include/beat_detector_hip.h."""

    _fields_ = [
        ("min_amplitude", C.c_uint32),
        ("min_beat_gap_ms", C.c_uint32),
        ("spectrum_max_hz", C.c_float),
        ("spectrum_threshold", C.c_float),
        ("spectrum_divisor", C.c_uint32),
        ("spectrum_include_dc", C.c_uint32),
        ("spectrum_inclusive", C.c_uint32),
        ("spectrum_dc_packs_nyquist", C.c_uint32),
        ("lpf_cutoff_hz", C.c_uint32),
        ("lpf_ratio", C.c_float),
    ]

    @classmethod
    def default(cls, **overrides) -> "Params":
        p = cls()
        lib().bd_default_params(C.byref(p))
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p


LIVE_ON_BEAT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)
ON_BEAT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32)
READ_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t)

FMT_I16, FMT_U16, FMT_F32 = 0, 1, 2
_FMT_DTYPE = {FMT_I16: np.int16, FMT_U16: np.uint16, FMT_F32: np.float32}


class OnsetParams(C.Structure):
    """bd_onset_params of include/beat_detector_hip.h (extension analysis)."""

    _fields_ = [
        ("fft_len", C.c_uint32),
        ("hop", C.c_uint32),
        ("window", C.c_uint32),
        ("band_max_hz", C.c_float),
        ("mean_frames", C.c_uint32),
        ("max_frames", C.c_uint32),
        ("delta", C.c_float),
        ("lambda_", C.c_float),
        ("min_gap_ms", C.c_uint32),
        ("use_median", C.c_uint32),
        ("flux_power", C.c_uint32),
    ]

    @classmethod
    def default(cls, **overrides) -> "OnsetParams":
        p = cls()
        lib().bd_onset_default_params(C.byref(p))
        if "fft_len" in overrides and "hop" not in overrides:
            overrides["hop"] = overrides["fft_len"] // 2
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p


class BiquadDesign(C.Structure):
    """bd_biquad_design of include/beat_detector_hip.h (extension analysis)."""

    _fields_ = [("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("na1", C.c_float), ("na2", C.c_float),
                ("c1b", C.c_float), ("c2b", C.c_float), ("pw", (C.c_float * 4) * 6), ("q", (C.c_float * 4) * 64)]


class BiquadParams(C.Structure):
    """bd_biquad_params of include/beat_detector_hip.h (extension analysis)."""

    _fields_ = [("cutoff_hz", C.c_uint32), ("q_factor", C.c_float), ("window_len", C.c_uint32),
                ("ratio", C.c_float), ("min_amplitude", C.c_uint32)]

    @classmethod
    def default(cls, **overrides) -> "BiquadParams":
        p = cls()
        lib().bd_biquad_default_params(C.byref(p))
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p


class InputSource(C.Structure):
    """bd_input_source: the stand-in for a cpal input device."""

    _fields_ = [
        ("sampling_rate", C.c_uint32),
        ("sample_format", C.c_int),
        ("max_callback_len", C.c_uint32),
        ("read", READ_FN),
        ("user", C.c_void_p),
    ]


_lib = None


def lib_path() -> str:
    # BEATDET_LIB selects another build of the same library (A/B experiments)
    return os.environ.get("BEATDET_LIB") or os.path.join(_HERE, _LIB_NAME)


def lib() -> C.CDLL:
    """Loads libbeatdet_hip.so (built in-tree by `__graft_entry__.build()` or
    `make -C beat_detector_amd/csrc`).  No fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise BeatDetectorError(BD_ERR_HIP, f"{path} is missing: build it with `make -C beat_detector_amd/csrc`")
    L = C.CDLL(path)
    u32, u64, i32, vp, f32 = C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_float
    pp = C.POINTER(Params)
    op = C.POINTER(OnsetParams)
    sigs = {
        "bd_default_params": (None, [pp]),
        "bd_strerror": (C.c_char_p, [i32]),
        "bd_last_error": (C.c_char_p, []),
        "bd_device_count": (i32, [C.POINTER(i32)]),
        "bd_device_name": (i32, [i32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]),
        "bd_current_device": (i32, [C.POINTER(i32)]),
        "bd_set_device": (i32, [i32]),
        "bd_kind_count": (i32, []),
        "bd_kind_name": (C.c_char_p, [i32]),
        "bd_kind_description": (C.c_char_p, [i32]),
        "bd_kind_min_duration_between_beats_ms": (u32, [i32]),
        "bd_detector_create": (i32, [i32, u32, pp, i32, C.POINTER(vp)]),
        "bd_detector_is_beat": (i32, [vp, vp, C.c_size_t, C.POINTER(u32)]),
        "bd_detector_kind": (i32, [vp]),
        "bd_detector_last": (i32, [vp, C.POINTER(f32), C.POINTER(i32)]),
        "bd_detector_destroy": (None, [vp]),
        "bd_batch_callbacks": (u64, [u64, u32]),
        "bd_batch_max_beats": (u32, [u32, u32, pp, u64, u32]),
        "bd_analyze_batch": (i32, [i32, u32, pp, vp, u64, u64, u64, u32, vp, vp, u64, vp, vp, vp, u32, vp]),
        "bd_analyze_batch_noalloc": (i32, [i32, u32, pp, vp, u64, u64, u64, u32, vp, vp, vp, u32, vp]),
        "bd_analyze_batch_ragged": (i32, [i32, u32, pp, vp, u64, vp, u64, u64, u32, vp, vp, u64, vp, vp, vp, u32, vp]),
        "bd_batch_candidates": (i32, [i32, u32, pp, vp, u64, u64, u64, u32, vp, vp, u64, vp]),
        "bd_analyze_batch_host": (i32, [i32, u32, pp, vp, u64, u64, u64, u32, vp, vp, vp, vp, vp, u32, i32]),
        "bd_analyze_batch_multi": (i32, [i32, u32, pp, vp, u64, u64, u64, u32, vp, vp, vp, vp, vp, u32, C.POINTER(i32), i32]),
        "bd_analyze_batch_devices": (i32, [i32, u32, pp, C.POINTER(vp), C.POINTER(u64), u64, u64, u32, C.POINTER(vp), C.POINTER(vp),
                                           u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u32, C.POINTER(i32), i32]),
        "bd_refractory": (i32, [u32, pp, u32, vp, u64, u64, u64, vp, vp, vp, u32, vp]),
        "bd_u16_data_to_i16": (i32, [vp, vp, u64, vp]),
        "bd_f32_data_to_i16": (i32, [vp, vp, u64, vp]),
        "bd_stereo_to_mono": (i32, [vp, vp, u64, vp]),
        "bd_synth_fill": (i32, [vp, u64, u64, u64, u64, u64, u32, vp]),
        "bd_synth_fill_host": (i32, [vp, u64, u64, u64, u64, u64, u32]),
        "bd_last_kernel_name": (C.c_char_p, []),
        "bd_tuning_reload": (None, []),
        "bd_live_create": (i32, [i32, u32, pp, u32, u32, u32, i32, C.POINTER(vp)]),
        "bd_live_set_on_beat": (i32, [vp, LIVE_ON_BEAT_FN, vp]),
        "bd_live_submit": (i32, [vp, vp, u32, i32, C.POINTER(u64)]),
        "bd_live_submit_ragged": (i32, [vp, vp, vp, u32, i32, C.POINTER(u64)]),
        "bd_live_poll": (i32, [vp, u64, i32, vp, vp, vp]),
        "bd_live_destroy": (None, [vp]),
        "bd_start_listening": (i32, [ON_BEAT_FN, vp, C.POINTER(InputSource), i32, C.POINTER(C.c_int), C.POINTER(vp)]),
        "bd_listener_join": (i32, [vp]),
        "bd_onset_default_params": (None, [op]),
        "bd_onset_frames": (u64, [u64, u32, u32]),
        "bd_onset_stft": (i32, [u32, op, vp, u64, u64, u64, vp, vp, u64, vp]),
        "bd_onset_pick": (i32, [u32, op, vp, u64, u64, u64, vp, vp, vp, vp, u32, vp]),
        "bd_onset_batch": (i32, [u32, op, vp, u64, u64, u64, vp, vp, u64, vp, vp, vp, u32, vp]),
        "bd_probe_read": (i32, [vp, u64, vp, u32, vp]),
        "bd_probe_read_ex": (i32, [vp, u64, vp, u32, u32, u32, vp]),
        "bd_chunk_source_synth": (i32, [vp, vp, u64, u64, u64, u64, vp]),
        "bd_chunked_create": (i32, [u64, u64, i32, C.POINTER(vp)]),
        "bd_chunked_device_bytes": (u64, [vp]),
        "bd_chunked_analyze": (i32, [vp, i32, u32, pp, u32, vp, vp, u64, vp, vp, vp, u32, vp]),
        "bd_chunked_onset": (i32, [vp, u32, op, vp, vp, u64, vp, vp, vp, u32, vp]),
        "bd_chunked_last_timing": (i32, [vp, vp, vp, u32]),
        "bd_chunked_destroy": (None, [vp]),
        "bd_biquad_default_params": (None, [C.POINTER(BiquadParams)]),
        "bd_biquad_design_lowpass": (i32, [u32, u32, f32, C.POINTER(BiquadDesign)]),
        "bd_biquad_envelope": (i32, [u32, C.POINTER(BiquadParams), C.POINTER(BiquadDesign), vp, u64, u64, u64, vp, vp,
                                     u64, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "bd_default_params bd_strerror bd_last_error bd_device_count bd_device_name bd_current_device bd_set_device bd_kind_count bd_kind_name "
    "bd_kind_description bd_kind_min_duration_between_beats_ms bd_detector_create "
    "bd_detector_is_beat bd_detector_kind bd_detector_last bd_detector_destroy bd_batch_callbacks "
    "bd_batch_max_beats bd_analyze_batch bd_analyze_batch_noalloc bd_analyze_batch_ragged bd_batch_candidates bd_analyze_batch_host "
    "bd_analyze_batch_multi bd_analyze_batch_devices bd_refractory bd_u16_data_to_i16 bd_f32_data_to_i16 bd_stereo_to_mono bd_synth_fill "
    "bd_synth_fill_host bd_last_kernel_name bd_tuning_reload bd_live_create bd_live_set_on_beat bd_live_submit bd_live_submit_ragged bd_live_poll "
    "bd_live_destroy bd_start_listening bd_listener_join bd_onset_default_params bd_onset_frames "
    "bd_onset_stft bd_onset_pick bd_onset_batch bd_biquad_default_params bd_biquad_design_lowpass bd_biquad_envelope bd_probe_read bd_probe_read_ex "
    "bd_chunk_source_synth bd_chunked_create bd_chunked_device_bytes bd_chunked_analyze bd_chunked_onset bd_chunked_last_timing bd_chunked_destroy"
).split()


def _check(status: int) -> int:
    if status < 0:
        raise BeatDetectorError(status, lib().bd_last_error().decode() or lib().bd_strerror(status).decode())
    return status


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().bd_device_count(C.byref(n)))
    return n.value


def device_name(device: int):
    """(name, total memory in bytes) of GPU `device` (bd_device_name)."""
    buf = C.create_string_buffer(256)
    mem = C.c_uint64(0)
    _check(lib().bd_device_name(int(device), buf, len(buf), C.byref(mem)))
    return buf.value.decode(), int(mem.value)


def last_kernel_name() -> str:
    return lib().bd_last_kernel_name().decode()


@contextlib.contextmanager
def tuning(**switches):
    """Sets the library's experiment switches (environment variables BD_*, named here in lower case
    without the prefix: tuning(spec_no_split=1)) for the calls inside the block.  The library reads
    its environment once; this publishes the changed environment with bd_tuning_reload and restores
    the previous one afterwards.  A value of None removes the variable.  Tests and experiment drivers only."""
    saved = {}
    for k, v in switches.items():
        name = "BD_" + k.upper()
        saved[name] = os.environ.get(name)
        if v is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(v)
    lib().bd_tuning_reload()
    try:
        yield
    finally:
        for name, v in saved.items():
            if v is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = v
        lib().bd_tuning_reload()


# --------------------------------------------------------------------------
# Reference interface mirror
# --------------------------------------------------------------------------
class BeatInfo:
    """src/lib.rs:51-66."""

    __slots__ = ("_relative_ms",)

    def __init__(self, relative_ms: int):
        self._relative_ms = int(relative_ms)

    def relative_ms(self) -> int:
        return self._relative_ms

    def __repr__(self) -> str:
        return f"BeatInfo {{ relative_ms: {self._relative_ms} }}"

    def __eq__(self, other) -> bool:
        return isinstance(other, BeatInfo) and other._relative_ms == self._relative_ms


class StrategyKind(enum.Enum):
    """src/lib.rs:142-185.  Values follow StrategyKind::values()."""

    LPF = 0
    Spectrum = 1

    def detector(self, sampling_rate: int, params: Optional[Params] = None, device: int = -1) -> "Strategy":
        """StrategyKind::detector, src/lib.rs:155-161."""
        return Strategy(self, sampling_rate, params, device)

    def name_(self) -> str:
        """StrategyKind::name, src/lib.rs:164-170 (`name` is taken by Enum)."""
        return lib().bd_kind_name(self.value).decode()

    def description(self) -> str:
        """StrategyKind::description, src/lib.rs:173-179."""
        return lib().bd_kind_description(self.value).decode()

    def min_duration_between_beats_ms(self) -> int:
        return lib().bd_kind_min_duration_between_beats_ms(self.value)

    @classmethod
    def values(cls) -> List["StrategyKind"]:
        """StrategyKind::values, src/lib.rs:182-184."""
        return [cls.LPF, cls.Spectrum]


class Strategy:
    """A detector (`Box<dyn Strategy + Send>`, src/lib.rs:72-137) whose state and
    arithmetic live on the GPU.  One thread at a time per object."""

    def __init__(self, kind: StrategyKind, sampling_rate: int, params: Optional[Params] = None, device: int = -1):
        self._h = C.c_void_p()
        self._kind = kind
        _check(lib().bd_detector_create(kind.value, sampling_rate, C.byref(params) if params is not None else None,
                                        device, C.byref(self._h)))

    def is_beat(self, samples) -> Optional[BeatInfo]:
        """Strategy::is_beat(&self, samples: &[i16]) -> Option<BeatInfo>, src/lib.rs:78."""
        a = np.ascontiguousarray(samples, dtype=np.int16)
        ms = C.c_uint32(0)
        r = _check(lib().bd_detector_is_beat(self._h, a.ctypes.data_as(C.c_void_p), a.size, C.byref(ms)))
        return BeatInfo(ms.value) if r == 1 else None

    def kind(self) -> StrategyKind:
        return StrategyKind(_check(lib().bd_detector_kind(self._h)))

    def last(self):
        """(analysed value, decision before the refractory rule) of the last call."""
        v, c = C.c_float(0), C.c_int(0)
        _check(lib().bd_detector_last(self._h, C.byref(v), C.byref(c)))
        return v.value, bool(c.value)

    def close(self) -> None:
        if self._h:
            lib().bd_detector_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------
# Batched entry points
# --------------------------------------------------------------------------
def batch_callbacks(n_samples: int, callback_len: int) -> int:
    return lib().bd_batch_callbacks(n_samples, callback_len)


def batch_max_beats(kind: StrategyKind, sampling_rate: int, n_samples: int, callback_len: int,
                    params: Optional[Params] = None) -> int:
    return lib().bd_batch_max_beats(kind.value, sampling_rate, C.byref(params) if params is not None else None,
                                    n_samples, callback_len)


def _stream_ptr(stream) -> Optional[int]:
    if stream is None:
        if torch is not None and torch.cuda.is_available():
            return torch.cuda.current_stream().cuda_stream
        return None
    return getattr(stream, "cuda_stream", stream)


def _pp(params):
    return C.byref(params) if params is not None else None


def _dev_ptr(t):
    return None if t is None else t.data_ptr()


def _require_cuda(t, dtype, name):
    if torch is None:
        raise BeatDetectorError(BD_ERR_NO_DEVICE, "torch is required for device tensors")
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise BeatDetectorError(BD_ERR_INVALID_ARG, f"{name} must be a contiguous CUDA tensor of {dtype}")


def _out_rows(n_streams: int, n_cb: int, dtype, device):
    """[n_streams, n_cb] view of rows padded to a multiple of 128 entries, so
    that every row starts on a 128-byte boundary (full-line stores)."""
    pitch = max(128, (n_cb + 127) & ~127)
    return torch.empty((n_streams, pitch), dtype=dtype, device=device)[:, :n_cb]


def _row_stride(flags, values=None) -> int:
    """Row pitch (in callbacks) shared by the flags and values tensors."""
    for t in (flags, values):
        if t is not None and (t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1)):
            raise BeatDetectorError(BD_ERR_INVALID_ARG, "outputs must be 2-D with unit inner stride")
    fs = flags.stride(0) if flags.shape[0] > 1 else max(flags.shape[1], 1)
    if values is not None and values.shape[0] > 1 and values.stride(0) != fs:
        raise BeatDetectorError(BD_ERR_INVALID_ARG, "values and flags must have the same row pitch")
    return fs


def batch_candidates(kind: StrategyKind, sampling_rate: int, samples, callback_len: int,
                     n_samples: Optional[int] = None, params: Optional[Params] = None,
                     want_values: bool = True, values=None, flags=None, stream=None):
    """First stage only: per-callback value and decision.  `samples` is a CUDA
    int16 tensor [n_streams, stride]."""
    _require_cuda(samples, torch.int16, "samples")
    n_streams, stride = samples.shape
    n = stride if n_samples is None else n_samples
    n_cb = n // callback_len if callback_len else 0
    if flags is None:
        flags = _out_rows(n_streams, n_cb, torch.uint8, samples.device)
    if values is None and want_values:
        values = _out_rows(n_streams, n_cb, torch.float32, samples.device)
    out_stride = _row_stride(flags, values)
    with torch.cuda.device(samples.device):
        _check(lib().bd_batch_candidates(kind.value, sampling_rate, _pp(params), samples.data_ptr(), n_streams, n,
                                         stride, callback_len, _dev_ptr(values), flags.data_ptr(), out_stride,
                                         _stream_ptr(stream)))
    return values, flags


def refractory(sampling_rate: int, callback_len: int, flags, max_beats: int, params: Optional[Params] = None,
               stream=None):
    if not (isinstance(flags, torch.Tensor) and flags.is_cuda and flags.dtype == torch.uint8):
        raise BeatDetectorError(BD_ERR_INVALID_ARG, "flags must be a CUDA uint8 tensor")
    n_streams, n_cb = flags.shape
    dev = flags.device
    count = torch.empty((n_streams,), dtype=torch.int32, device=dev)
    idx = torch.zeros((n_streams, max_beats), dtype=torch.int32, device=dev)
    ms = torch.zeros((n_streams, max_beats), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _check(lib().bd_refractory(sampling_rate, _pp(params), callback_len, flags.data_ptr(), _row_stride(flags),
                                   n_streams, n_cb,
                                   count.data_ptr(), idx.data_ptr(), ms.data_ptr(), max_beats, _stream_ptr(stream)))
    return count, idx, ms


class BatchResult:
    __slots__ = ("values", "flags", "beat_count", "beat_idx", "beat_ms")

    def __init__(self, values, flags, beat_count, beat_idx, beat_ms):
        self.values, self.flags = values, flags
        self.beat_count, self.beat_idx, self.beat_ms = beat_count, beat_idx, beat_ms

    def beats(self, stream_index: int):
        """[(callback index, relative_ms)] of one stream (host lists)."""
        n = int(self.beat_count[stream_index])
        n = min(n, self.beat_idx.shape[1])
        idx = np.asarray(self.beat_idx[stream_index][:n].cpu() if hasattr(self.beat_idx, "cpu") else self.beat_idx[stream_index][:n])
        ms = np.asarray(self.beat_ms[stream_index][:n].cpu() if hasattr(self.beat_ms, "cpu") else self.beat_ms[stream_index][:n])
        return list(zip(idx.astype(np.int64).tolist(), ms.astype(np.int64).tolist()))


def analyze_batch(kind: StrategyKind, sampling_rate: int, samples, callback_len: int,
                  n_samples: Optional[int] = None, params: Optional[Params] = None,
                  max_beats: Optional[int] = None, want_values: bool = True, out: Optional[BatchResult] = None,
                  stream=None) -> BatchResult:
    """bd_analyze_batch on a CUDA int16 tensor [n_streams, stride].  Enqueues on
    the current torch stream; results are CUDA tensors."""
    _require_cuda(samples, torch.int16, "samples")
    n_streams, stride = samples.shape
    n = stride if n_samples is None else n_samples
    n_cb = n // callback_len if callback_len else 0
    dev = samples.device
    if max_beats is None:
        max_beats = max(1, batch_max_beats(kind, sampling_rate, n, callback_len, params)) if callback_len else 1
    if out is None:
        out = BatchResult(
            _out_rows(n_streams, n_cb, torch.float32, dev) if want_values else None,
            _out_rows(n_streams, n_cb, torch.uint8, dev),
            torch.zeros((n_streams,), dtype=torch.int32, device=dev),
            torch.zeros((n_streams, max_beats), dtype=torch.int32, device=dev),
            torch.zeros((n_streams, max_beats), dtype=torch.int32, device=dev),
        )
    with torch.cuda.device(dev):
        _check(lib().bd_analyze_batch(kind.value, sampling_rate, _pp(params), samples.data_ptr(), n_streams, n, stride,
                                      callback_len, _dev_ptr(out.values), out.flags.data_ptr(),
                                      _row_stride(out.flags, out.values), out.beat_count.data_ptr(), out.beat_idx.data_ptr(), out.beat_ms.data_ptr(),
                                      out.beat_idx.shape[1], _stream_ptr(stream)))
    return out


def analyze_batch_devices(kind: StrategyKind, sampling_rate: int, shards, callback_len: int,
                          n_samples: Optional[int] = None, params: Optional[Params] = None,
                          max_beats: Optional[int] = None, want_values: bool = True) -> List[BatchResult]:
    """bd_analyze_batch_devices: `shards` is a list of CUDA int16 tensors [n_streams_g, stride] (the same stride),
    each resident on the GPU that is to analyse it; one host thread per shard inside the library, no inter-GPU
    traffic.  Returns one BatchResult per shard, on the shard's device; the call has finished when it returns."""
    if not shards:
        return []
    for t in shards:
        _require_cuda(t, torch.int16, "shard")
    stride = shards[0].shape[1]
    if any(t.shape[1] != stride for t in shards):
        raise BeatDetectorError(BD_ERR_INVALID_ARG, "all shards must have the same row stride")
    n = stride if n_samples is None else n_samples
    n_cb = n // callback_len if callback_len else 0
    if max_beats is None:
        max_beats = max(1, batch_max_beats(kind, sampling_rate, n, callback_len, params)) if callback_len else 1
    outs = []
    for t in shards:
        dev, rows = t.device, t.shape[0]
        outs.append(BatchResult(
            _out_rows(rows, n_cb, torch.float32, dev) if want_values else None,
            _out_rows(rows, n_cb, torch.uint8, dev),
            torch.zeros((rows,), dtype=torch.int32, device=dev),
            torch.zeros((rows, max_beats), dtype=torch.int32, device=dev),
            torch.zeros((rows, max_beats), dtype=torch.int32, device=dev)))
    for t in shards:
        torch.cuda.synchronize(t.device)            # the library's streams do not wait for torch's
    g = len(shards)
    arr = lambda ptrs: (C.c_void_p * g)(*ptrs)
    out_stride = _row_stride(outs[0].flags, outs[0].values)
    _check(lib().bd_analyze_batch_devices(
        kind.value, sampling_rate, _pp(params), arr([t.data_ptr() for t in shards]),
        (C.c_uint64 * g)(*[t.shape[0] for t in shards]), n, stride, callback_len,
        arr([o.values.data_ptr() for o in outs]) if want_values else None, arr([o.flags.data_ptr() for o in outs]), out_stride,
        arr([o.beat_count.data_ptr() for o in outs]), arr([o.beat_idx.data_ptr() for o in outs]),
        arr([o.beat_ms.data_ptr() for o in outs]), max_beats,
        (C.c_int * g)(*[t.device.index if t.device.index is not None else torch.cuda.current_device() for t in shards]), g))
    return outs


NOALLOC_MAX_CALLBACKS = 96 << 20  # BD_NOALLOC_SCRATCH_BYTES of include/beat_detector_hip.h


def analyze_batch_noalloc(kind: StrategyKind, sampling_rate: int, samples, callback_len: int, beat_count, beat_idx,
                          beat_ms, n_samples: Optional[int] = None, params: Optional[Params] = None,
                          stream=None) -> None:
    """bd_analyze_batch_noalloc: beat lists only, into tensors the caller already owns (CUDA int32
    [n_streams], [n_streams, max_beats] x 2); the per-callback decisions stay in the library's fixed
    device array, so nothing is allocated by the call."""
    _require_cuda(samples, torch.int16, "samples")
    _require_cuda(beat_count, torch.int32, "beat_count")
    _require_cuda(beat_idx, torch.int32, "beat_idx")
    _require_cuda(beat_ms, torch.int32, "beat_ms")
    n_streams, stride = samples.shape
    n = stride if n_samples is None else n_samples
    if beat_count.shape != (n_streams,) or beat_idx.shape != beat_ms.shape or beat_idx.shape[0] != n_streams:
        raise BeatDetectorError(BD_ERR_INVALID_ARG, "beat outputs must be [n_streams] and [n_streams, max_beats]")
    with torch.cuda.device(samples.device):
        _check(lib().bd_analyze_batch_noalloc(kind.value, sampling_rate, _pp(params), samples.data_ptr(), n_streams, n,
                                              stride, callback_len, beat_count.data_ptr(), beat_idx.data_ptr(),
                                              beat_ms.data_ptr(), beat_idx.shape[1], _stream_ptr(stream)))


def analyze_batch_ragged(kind: StrategyKind, sampling_rate: int, samples, lengths, callback_len: int,
                         params: Optional[Params] = None, max_beats: Optional[int] = None, want_values: bool = True,
                         stream=None) -> BatchResult:
    """bd_analyze_batch_ragged: row s of the CUDA int16 tensor holds lengths[s] samples (CUDA int64 tensor)."""
    _require_cuda(samples, torch.int16, "samples")
    _require_cuda(lengths, torch.int64, "lengths")
    n_streams, stride = samples.shape
    if lengths.shape != (n_streams,):
        raise BeatDetectorError(BD_ERR_INVALID_ARG, "lengths must have one entry per stream")
    n_cb = stride // callback_len if callback_len else 0
    dev = samples.device
    if max_beats is None:
        max_beats = max(1, batch_max_beats(kind, sampling_rate, stride, callback_len, params)) if callback_len else 1
    out = BatchResult(
        _out_rows(n_streams, n_cb, torch.float32, dev) if want_values else None,
        _out_rows(n_streams, n_cb, torch.uint8, dev),
        torch.zeros((n_streams,), dtype=torch.int32, device=dev),
        torch.zeros((n_streams, max_beats), dtype=torch.int32, device=dev),
        torch.zeros((n_streams, max_beats), dtype=torch.int32, device=dev),
    )
    with torch.cuda.device(dev):
        _check(lib().bd_analyze_batch_ragged(kind.value, sampling_rate, _pp(params), samples.data_ptr(), n_streams,
                                             lengths.data_ptr(), stride, stride, callback_len, _dev_ptr(out.values),
                                             out.flags.data_ptr(), _row_stride(out.flags, out.values),
                                             out.beat_count.data_ptr(), out.beat_idx.data_ptr(),
                                             out.beat_ms.data_ptr(), max_beats, _stream_ptr(stream)))
    return out


def _host_call(fn_name, kind, sampling_rate, samples, callback_len, params, max_beats, tail):
    a = np.ascontiguousarray(samples, dtype=np.int16)
    if a.ndim == 1:
        a = a[None, :]
    n_streams, n = a.shape
    n_cb = n // callback_len if callback_len else 0
    if max_beats is None:
        max_beats = max(1, batch_max_beats(kind, sampling_rate, n, callback_len, params)) if callback_len else 1
    values = np.zeros((n_streams, n_cb), dtype=np.float32)
    flags = np.zeros((n_streams, n_cb), dtype=np.uint8)
    count = np.zeros((n_streams,), dtype=np.uint32)
    idx = np.zeros((n_streams, max_beats), dtype=np.uint32)
    ms = np.zeros((n_streams, max_beats), dtype=np.uint32)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    _check(getattr(lib(), fn_name)(kind.value, sampling_rate, _pp(params), vp(a), n_streams, n, n, callback_len,
                                   vp(values), vp(flags), vp(count), vp(idx), vp(ms), max_beats, *tail))
    return BatchResult(values, flags, count, idx, ms)


def analyze_batch_host(kind: StrategyKind, sampling_rate: int, samples, callback_len: int,
                       params: Optional[Params] = None, max_beats: Optional[int] = None,
                       device: int = -1) -> BatchResult:
    """bd_analyze_batch_host on a numpy int16 array [n_streams, n_samples]."""
    return _host_call("bd_analyze_batch_host", kind, sampling_rate, samples, callback_len, params, max_beats, (device,))


def analyze_batch_multi(kind: StrategyKind, sampling_rate: int, samples, callback_len: int,
                        devices: Sequence[int], params: Optional[Params] = None,
                        max_beats: Optional[int] = None) -> BatchResult:
    """bd_analyze_batch_multi: contiguous stream shards over `devices`."""
    arr = (C.c_int * len(devices))(*devices)
    return _host_call("bd_analyze_batch_multi", kind, sampling_rate, samples, callback_len, params, max_beats,
                      (arr, len(devices)))


# --------------------------------------------------------------------------
# Ingest converters and the synthetic generator
# --------------------------------------------------------------------------
def u16_data_to_i16(data, stream=None):
    """record::u16_data_to_i16, src/record.rs:151-158, on a CUDA tensor.  torch
    has no uint16 arithmetic, so the input is an int16 tensor holding the u16 bits."""
    _require_cuda(data, torch.int16, "data")
    out = torch.empty_like(data)
    with torch.cuda.device(data.device):
        _check(lib().bd_u16_data_to_i16(data.data_ptr(), out.data_ptr(), data.numel(), _stream_ptr(stream)))
    return out


def f32_data_to_i16(data, stream=None):
    """record::f32_data_to_i16, src/record.rs:160-166, on a CUDA tensor."""
    _require_cuda(data, torch.float32, "data")
    out = torch.empty(data.shape, dtype=torch.int16, device=data.device)
    with torch.cuda.device(data.device):
        _check(lib().bd_f32_data_to_i16(data.data_ptr(), out.data_ptr(), data.numel(), _stream_ptr(stream)))
    return out


def stereo_to_mono(interleaved, stream=None):
    """Interleaved stereo CUDA int16 tensor [..., 2 * n] -> mono [..., n] (src/lib.rs:310-337)."""
    _require_cuda(interleaved, torch.int16, "interleaved")
    if interleaved.shape[-1] % 2:
        raise BeatDetectorError(BD_ERR_INVALID_ARG, "interleaved stereo needs an even number of samples")
    out = torch.empty(interleaved.shape[:-1] + (interleaved.shape[-1] // 2,), dtype=torch.int16, device=interleaved.device)
    with torch.cuda.device(interleaved.device):
        _check(lib().bd_stereo_to_mono(interleaved.data_ptr(), out.data_ptr(), out.numel(), _stream_ptr(stream)))
    return out


def synth_fill(samples, n_samples: Optional[int] = None, first_stream: int = 0, seed: int = 0x5EED,
               sampling_rate: int = 44100, stream=None):
    """Fills a CUDA int16 tensor [n_streams, stride] with synthetic click tracks."""
    _require_cuda(samples, torch.int16, "samples")
    n_streams, stride = samples.shape
    n = stride if n_samples is None else n_samples
    with torch.cuda.device(samples.device):
        _check(lib().bd_synth_fill(samples.data_ptr(), n_streams, n, stride, first_stream, seed, sampling_rate,
                                   _stream_ptr(stream)))
    return samples


def synth_fill_host(n_streams: int, n_samples: int, first_stream: int = 0, seed: int = 0x5EED,
                    sampling_rate: int = 44100) -> np.ndarray:
    """The same generator on the host (integer arithmetic only, bit-identical)."""
    out = np.zeros((n_streams, n_samples), dtype=np.int16)
    _check(lib().bd_synth_fill_host(out.ctypes.data_as(C.c_void_p), n_streams, n_samples, n_samples, first_stream,
                                    seed, sampling_rate))
    return out


# --------------------------------------------------------------------------
# Live path
# --------------------------------------------------------------------------
class Live:
    """bd_live: n_streams detectors advanced together, one callback per stream and
    tick, through a ring of pinned staging buffers and a HIP graph per tick."""

    def __init__(self, kind: StrategyKind, sampling_rate: int, n_streams: int, max_callback_len: int = 4096,
                 depth: int = 4, params: Optional[Params] = None, device: int = -1, on_beat=None):
        self._h = C.c_void_p()
        self.n_streams = n_streams
        _check(lib().bd_live_create(kind.value, sampling_rate, _pp(params), n_streams, max_callback_len, depth, device,
                                    C.byref(self._h)))
        self._cb = None
        if on_beat is not None:
            self._cb = LIVE_ON_BEAT_FN(lambda user, stream, ms: on_beat(stream, BeatInfo(ms)))
            _check(lib().bd_live_set_on_beat(self._h, self._cb, None))

    def submit(self, samples, fmt: int = FMT_I16) -> int:
        a = np.ascontiguousarray(samples, dtype=_FMT_DTYPE[fmt])
        if a.ndim != 2 or a.shape[0] != self.n_streams:
            raise BeatDetectorError(BD_ERR_INVALID_ARG, "samples must be [n_streams, n]")
        tick = C.c_uint64(0)
        _check(lib().bd_live_submit(self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], fmt, C.byref(tick)))
        return tick.value

    def submit_ragged(self, samples, lengths, fmt: int = FMT_I16) -> int:
        """One tick with a callback length per stream (bd_live_submit_ragged): `samples` is
        [n_streams, row_pitch], stream m's callback the first lengths[m] samples of row m; 0 skips the stream."""
        a = np.ascontiguousarray(samples, dtype=_FMT_DTYPE[fmt])
        n = np.ascontiguousarray(lengths, dtype=np.uint32)
        if a.ndim != 2 or a.shape[0] != self.n_streams or n.shape != (self.n_streams,):
            raise BeatDetectorError(BD_ERR_INVALID_ARG, "samples must be [n_streams, row_pitch], lengths [n_streams]")
        tick = C.c_uint64(0)
        _check(lib().bd_live_submit_ragged(self._h, a.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p),
                                           a.shape[1], fmt, C.byref(tick)))
        return tick.value

    def poll(self, tick: int, wait: bool = True):
        """None if the tick has not finished; else (beats u8[n], relative_ms u32[n], values f32[n])."""
        beats = np.zeros(self.n_streams, dtype=np.uint8)
        ms = np.zeros(self.n_streams, dtype=np.uint32)
        vals = np.zeros(self.n_streams, dtype=np.float32)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)
        r = _check(lib().bd_live_poll(self._h, tick, 1 if wait else 0, vp(beats), vp(ms), vp(vals)))
        return (beats, ms, vals) if r == 1 else None

    def close(self) -> None:
        if self._h:
            lib().bd_live_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Listener:
    """JoinHandle of start_listening."""

    def __init__(self, handle, keep):
        self._h, self._keep = handle, keep

    def join(self) -> None:
        if self._h:
            h, self._h = self._h, None
            _check(lib().bd_listener_join(h))


def start_listening(on_beat_cb, input_dev, strategy: StrategyKind, keep_recording) -> Listener:
    """record::start_listening(on_beat_cb, input_dev, strategy, keep_recording),
    src/record.rs:39-44.  `input_dev` stands in for the cpal device: an object
    with `sampling_rate`, `sample_format` (FMT_*), optional `max_callback_len`
    and `read(capacity) -> numpy array` (empty array = end of stream), or None.
    `keep_recording` is a ctypes.c_int used as the AtomicBool.  Raises
    BeatDetectorError where the reference returns Err(String)."""
    keep_refs = []
    src_ptr = None
    if input_dev is not None:
        fmt = int(input_dev.sample_format)
        dtype = _FMT_DTYPE[fmt]

        def _read(user, buf, capacity):
            data = np.ascontiguousarray(input_dev.read(int(capacity)), dtype=dtype)
            n = min(int(data.size), int(capacity))
            if n:
                C.memmove(buf, data.ctypes.data, n * data.itemsize)
            return n

        read_c = READ_FN(_read)
        src = InputSource(int(input_dev.sampling_rate), fmt, int(getattr(input_dev, "max_callback_len", 0)), read_c, None)
        keep_refs += [read_c, src, input_dev]
        src_ptr = C.byref(src)
    cb_c = ON_BEAT_FN(lambda user, ms: on_beat_cb(BeatInfo(ms)))
    keep_refs += [cb_c, keep_recording]
    h = C.c_void_p()
    _check(lib().bd_start_listening(cb_c, None, src_ptr, strategy.value, C.byref(keep_recording), C.byref(h)))
    return Listener(h, keep_refs)
