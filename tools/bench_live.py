#!/usr/bin/env python3
"""Latency of the live path (the live low-latency configuration): 256 concurrent streams,
1024-sample callbacks, one HIP graph replay per tick.  Prints one JSON line with
p50/p99 of submit -> result-on-host latency and the real-time margin."""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beat_detector_amd as bd

M = int(os.environ.get("LIVE_STREAMS", "256")); L = 1024; ticks = int(os.environ.get("LIVE_TICKS", "1000")); sr = 44100
out = {}
for kind in (bd.StrategyKind.Spectrum, bd.StrategyKind.LPF):
    x = bd.synth_fill_host(M, L * 64, seed=11)
    live = bd.Live(kind, sr, M, max_callback_len=L, depth=4)
    lat = []
    # the callbacks of a tick as the source delivers them: one contiguous [streams, L] block
    wins = [np.ascontiguousarray(x[:, i * L:(i + 1) * L]) for i in range(64)]
    gc.collect()
    gc.disable()      # a collection of the interpreter's heap inside a tick is not the path's latency
    for t in range(ticks + 20):
        w = wins[t % 64]
        t0 = time.perf_counter()
        tick = live.submit(w)
        live.poll(tick)
        dt = time.perf_counter() - t0
        if t >= 20:
            lat.append(dt * 1e6)
    gc.enable()
    live.close()
    lat = np.array(lat)
    top = np.argsort(lat)[-5:][::-1]
    print(kind.name, "slowest ticks (index after warm-up, us):", [(int(i), round(float(lat[i]), 1)) for i in top], file=sys.stderr)
    out[kind.name] = {"p50_us": float(np.percentile(lat, 50)), "p99_us": float(np.percentile(lat, 99)),
                      "max_us": float(lat.max()), "mean_us": float(lat.mean())}
print(json.dumps({"metric": "live callback-to-beat latency", "streams": M, "callback_len": L, "ticks": ticks,
                  "tick_period_us": L / sr * 1e6, "results": out}))
