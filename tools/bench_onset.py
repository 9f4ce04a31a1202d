"""Times bd_onset_stft / bd_onset_pick on synthetic tracks (quick experiment driver)."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import beat_detector_amd as bd

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=2048)
ap.add_argument("--seconds", type=float, default=300.0)
ap.add_argument("--fft", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
sr = 44100
n = int(a.seconds * sr)
stride = (n + 63) & ~63
dev = torch.empty((a.streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(dev, n_samples=n, sampling_rate=sr)
p = bd.OnsetParams.default(fft_len=a.fft)
out = bd.onset_batch(sr, dev, n_samples=n, params=p, max_onsets=4096)
torch.cuda.synchronize()
t_stft, t_pick = [], []
for _ in range(a.reps):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    bd.onset_stft(sr, dev, n_samples=n, params=p, flux=out.flux, band_energy=out.band_energy)
    e[1].record()
    bd.onset_pick(sr, out.flux, params=p, max_onsets=4096, want_candidates=False)
    e[2].record()
    torch.cuda.synchronize()
    t_stft.append(e[0].elapsed_time(e[1]))
    t_pick.append(e[1].elapsed_time(e[2]))
ms = min(t_stft)
print(json.dumps({"fft": a.fft, "streams": a.streams, "seconds": a.seconds, "stft_ms": t_stft, "pick_ms": t_pick,
                  "GBps": a.streams * n * 2 / ms / 1e6, "audio_s_per_s": a.streams * a.seconds / (ms + min(t_pick)) * 1e3,
                  "onsets_stream0": int(out.onset_count[0])}))
