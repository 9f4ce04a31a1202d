"""Experiment driver: what do the result stores of the half-block Spectrum kernel cost?

Runs the benchmark's batch (16,384 streams x 60 s, 512-sample callbacks) through the first-stage
kernel in one process, alternating the variants so that every variant sees the same placement of
the sample buffer.  With BEATDET_LIB=beat_detector_amd/libbeatdet_hip_exp_store.so the variants are
the store modes of that build (BD_SPEC_STORE):
  0 nt sc1 (the product's policy)   1 default   2 nt   3 sc1   4 nt sc0 sc1   5 sc0 sc1   6 sc0
  7 = mode 0 without the decisions either   8 = mode 0, values into one 1 MiB window
With BEATDET_LIB=...exp_run512.so the variants are run lengths (BD_SPEC_RUN 128 / 256 / 512).
Usage: exp_store.py modes 0,1,2 | runs 128,256,512 | grid 0:128,64000:128:3 (pad:run[:depth])      (environment: EXP_STREAMS, EXP_REPS, EXP_ROUNDS)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
from beat_detector_amd.api import _out_rows

streams = int(os.environ.get("EXP_STREAMS", "16384"))
reps = int(os.environ.get("EXP_REPS", "4"))
rounds = int(os.environ.get("EXP_ROUNDS", "3"))
sr, n, L = 44100, 44100 * 60, int(os.environ.get("EXP_L", "512"))
stride = (n + 63) & ~63
n_cb = n // L
K = bd.StrategyKind.Spectrum
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
vals = _out_rows(streams, n_cb, torch.float32, "cuda")
flags = _out_rows(streams, n_cb, torch.uint8, "cuda")
torch.cuda.synchronize()


def t(with_values):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bd.batch_candidates(K, sr, buf, L, n_samples=n, values=vals if with_values else None, flags=flags, want_values=with_values)
    a.record()
    for _ in range(reps):
        bd.batch_candidates(K, sr, buf, L, n_samples=n, values=vals if with_values else None, flags=flags, want_values=with_values)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


what, items = sys.argv[1], sys.argv[2].split(",")
print(f"lib {bd.lib_path()}  {streams} streams x 60 s, L = {L}, {reps} launches per measurement", flush=True)
for r in range(rounds):
    row = []
    for it in items:
        if what == "grid":      # pad:run pairs (unused dynamic LDS in bytes : half-blocks per wavefront task)
            pad, run, depth = (it.split(":") + ["0"])[:3]
            knobs = {"spec_lds_pad": pad, "spec_run": run, "spec_depth": depth}
        else:
            knobs = {{"modes": "spec_store", "runs": "spec_run"}[what]: it}
        with bd.tuning(**knobs):
            row.append(f"{what[:-1] if what != 'grid' else 'pad:run'} {it}: values {t(True):.3f} / none {t(False):.3f} ms")
    print(f"round {r}: " + " | ".join(row), flush=True)
