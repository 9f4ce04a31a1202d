import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
sr = 44100; streams = 4096; n = sr * 60; stride = (n + 63) & ~63
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
row = []
for L in [int(v) for v in os.environ.get("LENGTHS", "441,450,500,517,529,540,600,882,1000,480,1024").split(",")]:
    for _ in range(2):
        bd.batch_candidates(bd.StrategyKind.LPF, sr, buf, L, n_samples=n)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        bd.batch_candidates(bd.StrategyKind.LPF, sr, buf, L, n_samples=n)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    row.append((L, round(streams * (n // L) * L * 2 / ms / 1e9, 2)))
print(row)
