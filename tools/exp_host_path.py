"""PCIe-inclusive rate of the host-pointer entry point (pageable numpy memory)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beat_detector_amd as bd
sr = 44100
out = []
for streams, seconds in ((4096, 60),):
    n = sr * seconds
    x = bd.synth_fill_host(streams, n, seed=3)
    for thr in (2, 4, 8, 16, 32):
      os.environ['BD_HOST_THREADS'] = str(thr); bd.lib().bd_tuning_reload()
      for kind in (bd.StrategyKind.Spectrum,):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            res = bd.analyze_batch_host(kind, sr, x, 1024)
            ts.append(time.perf_counter() - t0)
        best = min(ts)
        out.append({"threads": thr, "kind": kind.name, "streams": streams, "GB": x.nbytes / 1e9, "s": round(best, 3),
                    "GBps": round(x.nbytes / best / 1e9, 2), "audio_s_per_s": round(streams * seconds / best),
                    "beats": int(np.sum(res.beat_count))})
        print(out[-1], flush=True)
