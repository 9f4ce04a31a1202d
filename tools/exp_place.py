"""Experiment driver: does the cost of the detection-value stores of the half-block Spectrum kernel depend on
where the value buffer lies?  One process, one sample buffer (the benchmark's batch), several value buffers:
separate allocations made before and after the sample buffer, and windows of one large allocation at
different offsets.  Prints, per buffer, its address and the launch time with values (and the time without)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd

streams = int(os.environ.get("EXP_STREAMS", "16384"))
reps = int(os.environ.get("EXP_REPS", "4"))
sr, n, L = 44100, 44100 * 60, 512
stride = (n + 63) & ~63
n_cb = n // L
out_stride = (n_cb + 127) & ~127
K = bd.StrategyKind.Spectrum
rows = streams * out_stride


def vbuf():
    return torch.empty((streams, out_stride), dtype=torch.float32, device="cuda")


early = [vbuf() for _ in range(2)]                 # before the samples
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
late = [vbuf() for _ in range(3)]                  # after the samples
slab = torch.empty((12 * rows + (1 << 20),), dtype=torch.float32, device="cuda")   # 4.1 GB, windows at odd offsets
wins = [slab[o:o + rows].view(streams, out_stride) for o in (0, rows + 32 * 1024, 5 * rows + 17 * 32, 11 * rows + 3 * 1024 * 1024 // 4)]
flags = torch.empty((streams, out_stride), dtype=torch.uint8, device="cuda")
flags2 = torch.empty((streams, out_stride), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def t(vals, fl=flags):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kw = dict(n_samples=n, values=vals[:, :n_cb] if vals is not None else None, flags=fl[:, :n_cb], want_values=vals is not None)
    bd.batch_candidates(K, sr, buf, L, **kw)
    a.record()
    for _ in range(reps):
        bd.batch_candidates(K, sr, buf, L, **kw)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print(f"samples at {buf.data_ptr():#x}; flags at {flags.data_ptr():#x}, {flags2.data_ptr():#x}", flush=True)
for r in range(2):
    print(f"round {r}: no values: {t(None):.3f} ms (flags 2: {t(None, flags2):.3f})", flush=True)
    for name, group in (("early", early), ("late", late), ("slab window", wins)):
        for v in group:
            print(f"  {name:12s} {v.data_ptr():#x} (mod 2 MiB {v.data_ptr() % (1 << 21):#x}): {t(v):.3f} ms", flush=True)
