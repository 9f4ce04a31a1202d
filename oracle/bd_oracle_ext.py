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
    ]


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
