"""Extension analysis: carried-state biquad envelope on the GPU.

No counterpart in the reference (its low-pass is first order and restarts in
every window, src/strategies/lpf.rs:66-71).  A second-order low-pass runs through
each whole stream; per decision window max |y| is compared with
trunc(ratio * max |x|).  Computed by ``libbeatdet_hip.so`` (`bd_biquad_*`); the
per-window decisions feed the same refractory pass as the strategies.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

from . import api
from .api import BiquadDesign, BiquadParams, _check, _dev_ptr, _out_rows, _require_cuda, _row_stride, _stream_ptr, lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def biquad_design(sampling_rate: int, cutoff_hz: int = 120, q_factor: float = 0.70710678) -> BiquadDesign:
    """Coefficients and chunk-transition tables (host only)."""
    d = BiquadDesign()
    _check(lib().bd_biquad_design_lowpass(sampling_rate, cutoff_hz, q_factor, C.byref(d)))
    return d


def biquad_envelope(sampling_rate: int, samples, n_samples: Optional[int] = None,
                    params: Optional[BiquadParams] = None, design: Optional[BiquadDesign] = None,
                    want_values: bool = True, values=None, flags=None, stream=None):
    """(values, flags) per decision window for a CUDA int16 tensor [n_streams, stride]."""
    _require_cuda(samples, torch.int16, "samples")
    p = params if params is not None else BiquadParams.default()
    n_streams, stride = samples.shape
    n = stride if n_samples is None else n_samples
    n_win = n // p.window_len if p.window_len else 0
    if flags is None:
        flags = _out_rows(n_streams, n_win, torch.uint8, samples.device)
    if values is None and want_values:
        values = _out_rows(n_streams, n_win, torch.float32, samples.device)
    out_stride = _row_stride(flags, values)
    with torch.cuda.device(samples.device):
        _check(lib().bd_biquad_envelope(sampling_rate, C.byref(p), C.byref(design) if design is not None else None,
                                        samples.data_ptr(), n_streams, n, stride, _dev_ptr(values), flags.data_ptr(),
                                        out_stride, _stream_ptr(stream)))
    return values, flags


def biquad_beats(sampling_rate: int, samples, n_samples: Optional[int] = None, params: Optional[BiquadParams] = None,
                 max_beats: Optional[int] = None, stream=None):
    """Envelope decisions followed by the refractory pass: (values, flags, beat_count, beat_idx, beat_ms)."""
    p = params if params is not None else BiquadParams.default()
    values, flags = biquad_envelope(sampling_rate, samples, n_samples, p, stream=stream)
    n = samples.shape[1] if n_samples is None else n_samples
    if max_beats is None:
        max_beats = max(1, api.batch_max_beats(api.StrategyKind.LPF, sampling_rate, n, p.window_len))
    count, idx, ms = api.refractory(sampling_rate, p.window_len, flags, max_beats, stream=stream)
    return values, flags, count, idx, ms
