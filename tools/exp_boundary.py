"""Experiment: how launch boundaries / interleaved kernels affect the Spectrum kernel's time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
from beat_detector_amd.api import _out_rows

streams = int(os.environ.get("EXP_STREAMS", "16384")); sr = 44100; n = sr * 60; L = 512
stride = (n + 63) & ~63
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
n_cb = n // L
vals = _out_rows(streams, n_cb, torch.float32, "cuda"); flags = _out_rows(streams, n_cb, torch.uint8, "cuda")
K = bd.StrategyKind.Spectrum
def s1(): bd.batch_candidates(K, sr, buf, L, n_samples=n, values=vals, flags=flags)
def s2(): bd.refractory(sr, L, flags, max_beats=141)
for _ in range(2): s1()
torch.cuda.synchronize()
def run(name, body, reps=6):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for e0, e1 in evs:
        body(e0, e1)
    b.record(); torch.cuda.synchronize()
    per = []
    for e0, e1 in evs:
        try:
            per.append(round(e0.elapsed_time(e1), 3))
        except Exception:
            pass
    print(f"{name}: total/rep {a.elapsed_time(b)/reps:.3f} ms; per-launch {per}", flush=True)
def plain(e0, e1): s1()
def with_events(e0, e1): e0.record(); s1(); e1.record()
def with_refr(e0, e1): e0.record(); s1(); e1.record(); s2()
def with_sync(e0, e1): e0.record(); s1(); e1.record(); torch.cuda.synchronize()
run("back-to-back", plain)
run("events around each", with_events)
run("events + refractory", with_refr)
run("events + host sync", with_sync)
