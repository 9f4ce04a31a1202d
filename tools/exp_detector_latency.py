"""Latency of one Strategy::is_beat call through the per-window detector handle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beat_detector_amd as bd
x = bd.synth_fill_host(1, 1024 * 2000, seed=2)[0]
out = {}
for kind in (bd.StrategyKind.Spectrum, bd.StrategyKind.LPF):
    det = kind.detector(44100)
    ts = []
    for j in range(2000):
        w = x[j * 1024:(j + 1) * 1024]
        t0 = time.perf_counter()
        det.is_beat(w)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[100:]) * 1e6
    out[kind.name] = {"p50_us": round(float(np.percentile(ts, 50)), 1), "p99_us": round(float(np.percentile(ts, 99)), 1)}
print(json.dumps(out))
