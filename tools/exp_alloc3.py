import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
from beat_detector_amd.api import _out_rows
streams = 16384; sr = 44100; n = sr * 60; L = 512; stride = (n + 63) & ~63
n_cb = n // L
K = bd.StrategyKind.Spectrum
out = []
def big():
    buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
    bd.synth_fill(buf, n_samples=n, seed=1)
    vals = _out_rows(streams, n_cb, torch.float32, "cuda") if os.environ.get("EXP_VALUES", "1") == "1" else None; flags = _out_rows(streams, n_cb, torch.uint8, "cuda")
    for _ in range(2): bd.batch_candidates(K, sr, buf, L, n_samples=n, values=vals, flags=flags, want_values=vals is not None)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): bd.batch_candidates(K, sr, buf, L, n_samples=n, values=vals, flags=flags, want_values=vals is not None)
    b.record(); torch.cuda.synchronize()
    out.append(f"B={a.elapsed_time(b)/3:.2f}")
    del buf, vals, flags; torch.cuda.empty_cache()
def small(gib, touch=True):
    d = torch.empty((gib << 29,), dtype=torch.int16, device="cuda")
    if touch: d.zero_(); torch.cuda.synchronize()
    del d; torch.cuda.empty_cache()
    out.append(f"s{gib}")
for tok in sys.argv[1:]:
    if tok == "B": big()
    elif tok.startswith("s"): small(int(tok[1:]))
    elif tok.startswith("u"): small(int(tok[1:]), touch=False)
print(" ".join(out), flush=True)
