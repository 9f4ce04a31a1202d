"""Spectrum throughput at sampling rates whose band up to 90 Hz holds more than three bins."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
streams = 2048
for sr in (44100, 24000, 22050, 16000, 11025, 8000):
    n = 44100 * 60; stride = (n + 63) & ~63
    buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
    bd.synth_fill(buf, n_samples=n, seed=1, sampling_rate=sr)
    row = []
    for env in ({}, {"BD_SPEC_NO_GRANULE": "1"}):
        os.environ.pop("BD_SPEC_NO_GRANULE", None); os.environ.update(env); bd.lib().bd_tuning_reload()
        for L in (512, 1024, 160, 441):
            for _ in range(2):
                bd.batch_candidates(bd.StrategyKind.Spectrum, sr, buf, L, n_samples=n)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                bd.batch_candidates(bd.StrategyKind.Spectrum, sr, buf, L, n_samples=n)
            b.record(); torch.cuda.synchronize()
            row.append((L, bd.last_kernel_name()[12:19], round(streams * n * 2 / (a.elapsed_time(b) / 3) / 1e6)))
    print(sr, row)
    del buf
