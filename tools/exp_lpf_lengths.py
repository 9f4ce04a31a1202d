"""LPF throughput for callback lengths off the 64-sample grid, per staging variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
sr = 44100; streams = 2048; n = sr * 60; stride = (n + 63) & ~63
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
for env in ({}, {"BD_LPF_VARIANT": "4"}, {"BD_LPF_LINES": "1"}, {"BD_LPF_LINES": "1", "BD_LPF_VARIANT": "4"}):
    for k in ("BD_LPF_VARIANT", "BD_LPF_LINES"):
        os.environ.pop(k, None)
    os.environ.update(env); bd.lib().bd_tuning_reload()
    row = []
    for L in (441, 882, 1000, 1500, 480, 160, 100, 1024):
        for _ in range(2):
            bd.batch_candidates(bd.StrategyKind.LPF, sr, buf, L, n_samples=n)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            bd.batch_candidates(bd.StrategyKind.LPF, sr, buf, L, n_samples=n)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        row.append((L, round(streams * n * 2 / ms / 1e6)))
    print(env, row)
