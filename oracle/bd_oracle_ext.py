"""ctypes binding of the CPU definition of the extension onset analysis
(oracle/bd_oracle_ext.c, part of oracle/libbd_oracle.so).

TEST INFRASTRUCTURE ONLY, and NO REFERENCE COUNTERPART: see bd_oracle_ext.h.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import bd_oracle


class OnsetParams(C.Structure):
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


class BiquadDesign(C.Structure):
    _fields_ = [("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float), ("na1", C.c_float), ("na2", C.c_float),
                ("c1b", C.c_float), ("c2b", C.c_float), ("pw", (C.c_float * 4) * 6), ("q", (C.c_float * 4) * 64)]


class BiquadParams(C.Structure):
    _fields_ = [("cutoff_hz", C.c_uint32), ("q_factor", C.c_float), ("window_len", C.c_uint32),
                ("ratio", C.c_float), ("min_amplitude", C.c_uint32)]


_ready = False


def lib() -> C.CDLL:
    global _ready
    L = bd_oracle.lib()
    if not _ready:
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        pp = C.POINTER(OnsetParams)
        L.bdo_onset_default_params.argtypes = [pp]
        L.bdo_onset_default_params.restype = None
        L.bdo_onset_frames.argtypes = [sz, u32, u32]
        L.bdo_onset_frames.restype = sz
        L.bdo_onset_frame_ms.argtypes = [sz, u32, u32, u32]
        L.bdo_onset_frame_ms.restype = u32
        L.bdo_ext_rfft_mag.argtypes = [vp, u32, u32, vp]
        L.bdo_ext_rfft_mag_f64.argtypes = [vp, u32, u32, vp]
        L.bdo_onset_stft.argtypes = [vp, sz, u32, pp, vp, vp]
        L.bdo_onset_pick.argtypes = [vp, sz, u32, pp, vp, vp, vp, sz]
        L.bdo_onset_pick.restype = sz
        L.bdo_biquad_default_params.argtypes = [C.POINTER(BiquadParams)]
        L.bdo_biquad_default_params.restype = None
        L.bdo_biquad_design_lowpass.argtypes = [u32, u32, C.c_float, C.POINTER(BiquadDesign)]
        L.bdo_biquad_envelope.argtypes = [vp, sz, C.POINTER(BiquadDesign), C.POINTER(BiquadParams), vp, vp, vp]
        L.bdo_biquad_filter_f64.argtypes = [vp, sz, u32, u32, C.c_double, vp]
        L.bdo_biquad_filter_f64.restype = None
        _ready = True
    return L


def default_params(**overrides) -> OnsetParams:
    p = OnsetParams()
    lib().bdo_onset_default_params(C.byref(p))
    if "fft_len" in overrides and "hop" not in overrides:
        overrides["hop"] = overrides["fft_len"] // 2
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def frames(n_samples: int, fft_len: int, hop: int) -> int:
    return lib().bdo_onset_frames(n_samples, fft_len, hop)


def frame_ms(j: int, fft_len: int, hop: int, sampling_rate: int) -> int:
    return lib().bdo_onset_frame_ms(j, fft_len, hop, sampling_rate)


def rfft_mag(frame, window: bool = True) -> np.ndarray:
    x = np.ascontiguousarray(frame, dtype=np.int16)
    out = np.empty(x.size // 2 + 1, dtype=np.float32)
    if lib().bdo_ext_rfft_mag(_vp(x), x.size, int(window), _vp(out)) != 0:
        raise ValueError("frame length must be 1024 or 2048")
    return out


def rfft_mag_f64(frame, window: bool = True) -> np.ndarray:
    x = np.ascontiguousarray(frame, dtype=np.int16)
    out = np.empty(x.size // 2 + 1, dtype=np.float64)
    if lib().bdo_ext_rfft_mag_f64(_vp(x), x.size, int(window), _vp(out)) != 0:
        raise ValueError("frame length must be 1024 or 2048")
    return out


def onset_stft(samples, sampling_rate: int, params: OnsetParams | None = None):
    """(flux, band_energy) of one stream."""
    p = params if params is not None else default_params()
    x = np.ascontiguousarray(samples, dtype=np.int16)
    nf = frames(x.size, p.fft_len, p.hop)
    flux = np.zeros(nf, dtype=np.float32)
    band = np.zeros(nf, dtype=np.float32)
    if lib().bdo_onset_stft(_vp(x), x.size, sampling_rate, C.byref(p), _vp(flux), _vp(band)) != 0:
        raise ValueError("unsupported onset parameters")
    return flux, band


def onset_pick(flux, sampling_rate: int, params: OnsetParams | None = None):
    """(candidates u8[n], onset_frame, onset_ms) of one flux series."""
    p = params if params is not None else default_params()
    f = np.ascontiguousarray(flux, dtype=np.float32)
    n = f.size
    cand = np.zeros(n, dtype=np.uint8)
    frame = np.zeros(max(n, 1), dtype=np.uint32)
    ms = np.zeros(max(n, 1), dtype=np.uint32)
    cnt = lib().bdo_onset_pick(_vp(f), n, sampling_rate, C.byref(p), _vp(cand), _vp(frame), _vp(ms), n)
    return cand, frame[:cnt].copy(), ms[:cnt].copy()


def biquad_params(**overrides) -> BiquadParams:
    p = BiquadParams()
    lib().bdo_biquad_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def biquad_design(sampling_rate: int, cutoff_hz: int = 120, q_factor: float = 0.70710678) -> BiquadDesign:
    d = BiquadDesign()
    if lib().bdo_biquad_design_lowpass(sampling_rate, cutoff_hz, q_factor, C.byref(d)) != 0:
        raise ValueError("unusable biquad parameters")
    return d


def biquad_envelope(samples, sampling_rate: int, params: BiquadParams | None = None, design: BiquadDesign | None = None,
                    want_y: bool = False):
    """(values f32[n_windows], flags u8[n_windows][, y f32[n_windows * window_len]]) of one stream."""
    p = params if params is not None else biquad_params()
    d = design if design is not None else biquad_design(sampling_rate, p.cutoff_hz, p.q_factor)
    x = np.ascontiguousarray(samples, dtype=np.int16)
    nw = x.size // p.window_len
    values = np.zeros(nw, dtype=np.float32)
    flags = np.zeros(nw, dtype=np.uint8)
    y = np.zeros(x.size, dtype=np.float32) if want_y else None
    rc = lib().bdo_biquad_envelope(_vp(x), x.size, C.byref(d), C.byref(p), _vp(values), _vp(flags),
                                   _vp(y) if want_y else None)
    if rc != 0:
        raise ValueError("window_len must be a power of two between 64 and 4096")
    return (values, flags, y[:nw * p.window_len]) if want_y else (values, flags)


def biquad_filter_f64(samples, sampling_rate: int, cutoff_hz: int = 120, q_factor: float = 0.70710678) -> np.ndarray:
    x = np.ascontiguousarray(samples, dtype=np.int16)
    y = np.zeros(x.size, dtype=np.float64)
    lib().bdo_biquad_filter_f64(_vp(x), x.size, sampling_rate, cutoff_hz, q_factor, _vp(y))
    return y
