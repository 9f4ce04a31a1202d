"""Extension analysis: STFT onset detection on the GPU.

No counterpart in the reference (its Spectrum strategy keeps three raw FFT bins
of an unwindowed ring and compares their maximum with a constant,
src/strategies/spectrum.rs:82-92).  Hann STFT of 1024 or 2048 points with
hop = half or a quarter of the length, low-band energy, spectral flux and adaptive-threshold
peak picking; all computed by ``libbeatdet_hip.so`` (`bd_onset_*`).
"""
from __future__ import annotations

from typing import Optional

from . import api
from .api import OnsetParams, _check, _dev_ptr, _require_cuda, _stream_ptr, lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

import ctypes as C


def onset_frames(n_samples: int, fft_len: int = 2048, hop: Optional[int] = None) -> int:
    return lib().bd_onset_frames(n_samples, fft_len, fft_len // 2 if hop is None else hop)


def frame_ms(j: int, fft_len: int, hop: int, sampling_rate: int) -> int:
    """Time stamp of frame j (its centre), integer arithmetic as in bd_onset_pick."""
    return ((j * hop + fft_len // 2) * 1000) // sampling_rate


def _rows(n_streams: int, n: int, dtype, device):
    pitch = max(128, (n + 127) & ~127)
    return torch.empty((n_streams, pitch), dtype=dtype, device=device)[:, :n]


class OnsetResult:
    __slots__ = ("flux", "band_energy", "candidates", "onset_count", "onset_frame", "onset_ms")

    def __init__(self, flux=None, band_energy=None, candidates=None, onset_count=None, onset_frame=None,
                 onset_ms=None):
        self.flux, self.band_energy, self.candidates = flux, band_energy, candidates
        self.onset_count, self.onset_frame, self.onset_ms = onset_count, onset_frame, onset_ms

    def onsets(self, stream_index: int):
        """[(frame, ms)] of one stream."""
        n = min(int(self.onset_count[stream_index]), self.onset_frame.shape[1])
        return list(zip(self.onset_frame[stream_index, :n].cpu().tolist(),
                        self.onset_ms[stream_index, :n].cpu().tolist()))


def onset_stft(sampling_rate: int, samples, n_samples: Optional[int] = None, params: Optional[OnsetParams] = None,
               want_band_energy: bool = True, flux=None, band_energy=None, stream=None):
    """flux and band energy per frame for a CUDA int16 tensor [n_streams, stride]."""
    _require_cuda(samples, torch.int16, "samples")
    p = params if params is not None else OnsetParams.default()
    n_streams, stride = samples.shape
    n = stride if n_samples is None else n_samples
    nf = onset_frames(n, p.fft_len, p.hop)
    if flux is None:
        flux = _rows(n_streams, nf, torch.float32, samples.device)
    if band_energy is None and want_band_energy:
        band_energy = _rows(n_streams, nf, torch.float32, samples.device)
    out_stride = flux.stride(0) if n_streams > 1 else max(flux.shape[1], 1)
    if band_energy is not None and n_streams > 1 and band_energy.stride(0) != out_stride:
        raise api.BeatDetectorError(api.BD_ERR_INVALID_ARG, "flux and band_energy must share their row pitch")
    with torch.cuda.device(samples.device):
        _check(lib().bd_onset_stft(sampling_rate, C.byref(p), samples.data_ptr(), n_streams, n, stride,
                                   flux.data_ptr(), _dev_ptr(band_energy), out_stride, _stream_ptr(stream)))
    return flux, band_energy


def onset_pick(sampling_rate: int, flux, params: Optional[OnsetParams] = None, max_onsets: Optional[int] = None,
               want_candidates: bool = True, stream=None) -> OnsetResult:
    """Peak picking over flux rows (CUDA float32 [n_streams, n_frames], unit inner stride)."""
    if not (isinstance(flux, torch.Tensor) and flux.is_cuda and flux.dtype == torch.float32 and flux.dim() == 2):
        raise api.BeatDetectorError(api.BD_ERR_INVALID_ARG, "flux must be a 2-D CUDA float32 tensor")
    p = params if params is not None else OnsetParams.default()
    n_streams, nf = flux.shape
    dev = flux.device
    stride = flux.stride(0) if n_streams > 1 else max(nf, 1)
    if max_onsets is None:
        max_onsets = max(1, nf)
    cand = torch.empty((n_streams, stride), dtype=torch.uint8, device=dev)[:, :nf] if want_candidates else None
    count = torch.zeros((n_streams,), dtype=torch.int32, device=dev)
    frame = torch.zeros((n_streams, max_onsets), dtype=torch.int32, device=dev)
    ms = torch.zeros((n_streams, max_onsets), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _check(lib().bd_onset_pick(sampling_rate, C.byref(p), flux.data_ptr(), n_streams, nf, stride,
                                   _dev_ptr(cand), count.data_ptr(), frame.data_ptr(), ms.data_ptr(), max_onsets,
                                   _stream_ptr(stream)))
    return OnsetResult(flux, None, cand, count, frame, ms)


def onset_batch(sampling_rate: int, samples, n_samples: Optional[int] = None, params: Optional[OnsetParams] = None,
                max_onsets: Optional[int] = None, want_band_energy: bool = True, out: Optional[OnsetResult] = None,
                stream=None) -> OnsetResult:
    """bd_onset_batch: STFT + peak picking in one call."""
    _require_cuda(samples, torch.int16, "samples")
    p = params if params is not None else OnsetParams.default()
    n_streams, stride = samples.shape
    n = stride if n_samples is None else n_samples
    nf = onset_frames(n, p.fft_len, p.hop)
    dev = samples.device
    if out is None:
        if max_onsets is None:
            max_onsets = max(1, nf)
        out = OnsetResult(
            _rows(n_streams, nf, torch.float32, dev),
            _rows(n_streams, nf, torch.float32, dev) if want_band_energy else None,
            None,
            torch.zeros((n_streams,), dtype=torch.int32, device=dev),
            torch.zeros((n_streams, max_onsets), dtype=torch.int32, device=dev),
            torch.zeros((n_streams, max_onsets), dtype=torch.int32, device=dev),
        )
    out_stride = out.flux.stride(0) if n_streams > 1 else max(out.flux.shape[1], 1)
    with torch.cuda.device(dev):
        _check(lib().bd_onset_batch(sampling_rate, C.byref(p), samples.data_ptr(), n_streams, n, stride,
                                    out.flux.data_ptr(), _dev_ptr(out.band_energy), out_stride,
                                    out.onset_count.data_ptr(), out.onset_frame.data_ptr(), out.onset_ms.data_ptr(),
                                    out.onset_frame.shape[1], _stream_ptr(stream)))
    return out
