"""Rates of the Spectrum kernels for callback lengths around 400-640 (tools, not product).
Environment: STREAMS, KIND (Spectrum | LPF), SR, LENGTHS (comma separated), KNOB=<switch>:<v1>,<v2>,... sweeps one experiment switch of the
library (for example KNOB=split_subruns:1,4,8) inside one process, so every value sees the same buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
SR = int(os.environ.get("SR", "44100"))     # sampling rate: decides how many bins the band holds
KIND = bd.StrategyKind[os.environ.get("KIND", "Spectrum")]
streams, n = int(os.environ.get("STREAMS", "2048")), SR * 60
buf = torch.empty((streams, (n + 63) & ~63), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=3)
knob, values = None, [None]
if os.environ.get("KNOB"):
    knob, v = os.environ["KNOB"].split(":")
    values = v.split(",")
for L in [int(v) for v in os.environ.get("LENGTHS", "441,450,500,504,517,529,540,600,640").split(",")]:
    n_cb = n // L
    row = []
    for val in values:
        with bd.tuning(**({knob: val} if knob else {})):
            res = bd.analyze_batch(KIND, SR, buf, L, n_samples=n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                bd.analyze_batch(KIND, SR, buf, L, n_samples=n, out=res)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
        row.append(f"{knob}={val} " * bool(knob) + f"{dt*1e3:7.3f} ms {streams*n_cb*L*2/dt/1e12:.2f} TB/s")
    print(f"L={L:5d} {bd.last_kernel_name():28s} " + " | ".join(row) + f"  beats {int(res.beat_count.sum())}", flush=True)
