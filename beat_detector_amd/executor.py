"""Chunked executor (bd_chunked_* of include/beat_detector_hip.h): one GPU works
through a shard of streams that is larger than its memory, a chunk of streams at
a time, the next chunk being produced on a side stream while the current one is
analysed.  Streams are independent in the reference (one detector state per
stream: src/strategies/mod.rs:39-56, src/strategies/spectrum.rs:38-44), so the
results equal those of one launch over the whole shard."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

from . import api
from .api import BD_ERR_INVALID_ARG, BeatDetectorError, OnsetParams, Params, StrategyKind, _check, _pp, _stream_ptr, lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

CHUNK_SOURCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                              C.c_void_p)


class SynthSource(C.Structure):
    """bd_synth_source: the synthetic click-track generator as a chunk source."""

    _fields_ = [("seed", C.c_uint64), ("first_stream", C.c_uint64), ("sampling_rate", C.c_uint32)]


class ChunkedExecutor:
    """Owns two device buffers of `chunk_streams` x `n_samples` samples."""

    def __init__(self, chunk_streams: int, n_samples: int, device: int = -1):
        self._h = C.c_void_p()
        self.chunk_streams, self.n_samples = chunk_streams, n_samples
        _check(lib().bd_chunked_create(chunk_streams, n_samples, device, C.byref(self._h)))
        self._keep = []

    def device_bytes(self) -> int:
        return lib().bd_chunked_device_bytes(self._h)

    def _source(self, source, first_stream, seed, sampling_rate):
        """(function pointer, user pointer) of a source: None = the synthetic generator;
        a Python callable `f(dst_ptr, first, rows, n_samples, stride, hip_stream) -> int` otherwise."""
        if source is None:
            user = SynthSource(seed, first_stream, sampling_rate)
            fn = C.cast(lib().bd_chunk_source_synth, CHUNK_SOURCE_FN)
            self._keep = [user, fn]
            return fn, C.cast(C.pointer(user), C.c_void_p)

        def _call(user, dst, first, rows, n, stride, st):
            try:
                return int(source(dst, first, rows, n, stride, st) or 0)
            except Exception:  # an exception must not cross the C boundary
                return BD_ERR_INVALID_ARG

        fn = CHUNK_SOURCE_FN(_call)
        self._keep = [fn, source]
        return fn, None

    def analyze(self, kind: StrategyKind, sampling_rate: int, callback_len: int, n_streams: int, beat_count, beat_idx,
                beat_ms, source: Optional[Callable] = None, first_stream: int = 0, seed: int = 0x5EED,
                params: Optional[Params] = None, stream=None) -> None:
        """bd_chunked_analyze into CUDA int32 tensors [n_streams], [n_streams, max_beats] x 2."""
        fn, user = self._source(source, first_stream, seed, sampling_rate)
        _check(lib().bd_chunked_analyze(self._h, kind.value, sampling_rate, _pp(params), callback_len, fn, user,
                                        n_streams, beat_count.data_ptr(), api._dev_ptr(beat_idx), api._dev_ptr(beat_ms),
                                        beat_idx.shape[1] if beat_idx is not None else 0, _stream_ptr(stream)))

    def onset(self, sampling_rate: int, n_streams: int, onset_count, onset_frame, onset_ms,
              params: Optional[OnsetParams] = None, source: Optional[Callable] = None, first_stream: int = 0,
              seed: int = 0x5EED, stream=None) -> None:
        """bd_chunked_onset (extension analysis) into CUDA int32 tensors."""
        p = params if params is not None else OnsetParams.default()
        fn, user = self._source(source, first_stream, seed, sampling_rate)
        _check(lib().bd_chunked_onset(self._h, sampling_rate, C.byref(p), fn, user, n_streams, onset_count.data_ptr(),
                                      api._dev_ptr(onset_frame), api._dev_ptr(onset_ms),
                                      onset_frame.shape[1] if onset_frame is not None else 0, _stream_ptr(stream)))

    def last_timing(self, capacity: int = 64):
        """([ms of the dominant kernel], [ms of the pass after it]) per chunk of the last pass (waits for it)."""
        a = (C.c_float * capacity)()
        b = (C.c_float * capacity)()
        n = _check(lib().bd_chunked_last_timing(self._h, a, b, capacity))
        n = min(n, capacity)
        return list(a[:n]), list(b[:n])

    def close(self) -> None:
        if self._h:
            lib().bd_chunked_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
