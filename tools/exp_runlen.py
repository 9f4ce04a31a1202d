import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
from beat_detector_amd.api import _out_rows
streams = 16384; sr = 44100; n = sr * 60; L = 512; stride = (n + 63) & ~63
n_cb = n // L
K = bd.StrategyKind.Spectrum
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
vals = _out_rows(streams, n_cb, torch.float32, "cuda"); flags = _out_rows(streams, n_cb, torch.uint8, "cuda")
def t(v):
    for _ in range(2): bd.batch_candidates(K, sr, buf, L, n_samples=n, values=v, flags=flags, want_values=v is not None)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): bd.batch_candidates(K, sr, buf, L, n_samples=n, values=v, flags=flags, want_values=v is not None)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 3, 2)
out = []
for run in (128, 64, 256, 512, 128):
    os.environ["BD_SPEC_RUN"] = str(run); bd.lib().bd_tuning_reload()
    out.append((run, t(vals), t(None)))
print("run, with values, flags only:", out, flush=True)
