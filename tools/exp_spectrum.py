"""Experiment driver: times the first-stage kernel for a few configurations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd

streams = int(os.environ.get("EXP_STREAMS", "8192"))
sr, n = 44100, 44100 * 60
stride = (n + 63) & ~63 if os.environ.get("EXP_ALIGN", "1") == "1" else n
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
torch.cuda.synchronize()

def timeit(kind, L, want_values=True, reps=5):
    n_cb = n // L
    from beat_detector_amd.api import _out_rows
    vals = _out_rows(streams, n_cb, torch.float32, "cuda") if want_values else None
    flags = _out_rows(streams, n_cb, torch.uint8, "cuda")
    for _ in range(2):
        bd.batch_candidates(kind, sr, buf, L, n_samples=n, values=vals, flags=flags, want_values=want_values)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        bd.batch_candidates(kind, sr, buf, L, n_samples=n, values=vals, flags=flags, want_values=want_values)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = streams * n_cb * L * 2 / 1e9
    return ms, gb / ms

which = sys.argv[1] if len(sys.argv) > 1 else "spectrum"
only = os.environ.get("EXP_ONLY")  # e.g. "512,1"
if which == "spectrum":
    for L in (512, 1024):
        for wv in (True, False):
            if only and only != f"{L},{int(wv)}":
                continue
            ms, tbps = timeit(bd.StrategyKind.Spectrum, L, wv)
            print(f"spectrum L={L} values={wv} run={os.environ.get('BD_SPEC_RUN','auto')}: {ms:.3f} ms  {tbps:.3f} TB/s", flush=True)
else:
    ms, tbps = timeit(bd.StrategyKind.LPF, 1024)
    print(f"lpf variant={os.environ.get('BD_LPF_VARIANT','0')}: {ms:.3f} ms {tbps:.3f} TB/s", flush=True)
