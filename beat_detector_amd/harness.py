"""The reference's offline harness (src/lib.rs:198-337 of the reference: decode
to mono i16, cut into windows, run every strategy) over the batched GPU entry
point.  Decoding: WAV / raw PCM (there is no MP3 decoder in this environment;
the reference uses minimp3 in its tests only)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .api import BeatInfo, StrategyKind, analyze_batch_host


def stereo_to_mono(interleaved) -> np.ndarray:
    """lib.rs:319-324: ((l as i32 + r as i32) as f32 / 2.0) as i16 -- truncates toward zero."""
    a = np.asarray(interleaved, dtype=np.int16).reshape(-1, 2).astype(np.int32)
    s = (a[:, 0] + a[:, 1]).astype(np.float32) / np.float32(2.0)
    return np.clip(np.trunc(s), -32768, 32767).astype(np.int16)


def read_wav_to_mono(path: str) -> Tuple[np.ndarray, int]:
    """Counterpart of read_mp3_to_mono (lib.rs:300-337) for 16-bit PCM WAV files."""
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if data.dtype != np.int16:
        raise ValueError("only 16-bit PCM WAV is supported")
    if data.ndim == 1:
        return data.copy(), int(rate)
    if data.shape[1] == 2:
        return stereo_to_mono(data.reshape(-1)), int(rate)
    raise ValueError(f"Unsupported number of channels={data.shape[1]}")  # lib.rs:328


def pad_like_the_reference(samples, window_length: int) -> np.ndarray:
    """lib.rs:270-274: appends `len % window_length` zeros (not the complement)."""
    a = np.asarray(samples, dtype=np.int16)
    remainder = a.size % window_length
    return np.concatenate([a, np.zeros(remainder, dtype=np.int16)]) if remainder else a


def apply_samples_to_all_strategies(window_length: int, samples, sampling_rate: int = 44100,
                                    device: int = -1) -> Dict[StrategyKind, List[BeatInfo]]:
    """lib.rs:263-297.  Note that the reference's harness always builds its
    detectors for 44100 Hz (lib.rs:284) whatever the file's rate is; pass the
    rate you want analysed."""
    padded = pad_like_the_reference(samples, window_length)
    out: Dict[StrategyKind, List[BeatInfo]] = {}
    for kind in (StrategyKind.LPF, StrategyKind.Spectrum):           # lib.rs:279
        res = analyze_batch_host(kind, sampling_rate, padded[None, :], window_length, device=device)
        out[kind] = [BeatInfo(ms) for _, ms in res.beats(0)]
    return out
