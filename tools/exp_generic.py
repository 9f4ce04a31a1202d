"""Throughput of the batched entry for callback lengths on and off the fast paths."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
sr = 44100; streams = 2048; n = sr * 60; stride = (n + 63) & ~63
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
out = []
for kind in (bd.StrategyKind.Spectrum, bd.StrategyKind.LPF):
    for L in (512, 1024, 256, 2048, 4096, 441, 480, 882, 1000, 1500, 160, 240, 720, 96, 48, 128, 64):
        for _ in range(2):
            bd.batch_candidates(kind, sr, buf, L, n_samples=n)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            bd.batch_candidates(kind, sr, buf, L, n_samples=n)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        out.append((kind.name, L, round(ms, 2), round(streams * n * 2 / ms / 1e6), bd.last_kernel_name()))
for o in out:
    print(o)
if os.environ.get("BD_AB_SMALL"):
    os.environ["BD_SPEC_NO_SMALL_GRANULE"] = "1"; bd.lib().bd_tuning_reload()
    for L in (480, 160, 240, 720, 96, 48):
        for _ in range(2):
            bd.batch_candidates(bd.StrategyKind.Spectrum, sr, buf, L, n_samples=n)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            bd.batch_candidates(bd.StrategyKind.Spectrum, sr, buf, L, n_samples=n)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        print(("window instead", L, round(ms, 2), round(streams * n * 2 / ms / 1e6), bd.last_kernel_name()))
