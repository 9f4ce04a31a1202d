"""Time of the refractory pass on the decisions of the headline batch (tools, not product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
SR, streams, n, L = 44100, int(os.environ.get("STREAMS", "16384")), 44100 * 60, int(os.environ.get("L", "512"))
buf = torch.empty((streams, (n + 63) & ~63), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=0x5EED)
res = bd.analyze_batch(bd.StrategyKind.Spectrum, SR, buf, L, n_samples=n, want_values=False)
torch.cuda.synchronize()
flags = res.flags
print("decisions set per stream:", float(flags.sum()) / streams, "beats per stream:", float(res.beat_count.sum()) / streams)
mb = res.beat_idx.shape[1]
L_ = bd.lib()
count, idx, ms = bd.refractory(SR, L, flags, max_beats=mb)
stride = flags.stride(0)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fl, with_lists=True, reps=20):
    args = (idx.data_ptr(), ms.data_ptr(), mb) if with_lists else (None, None, 0)
    L_.bd_refractory(SR, None, L, fl.data_ptr(), fl.stride(0), streams, fl.shape[1], count.data_ptr(), *args, None)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        L_.bd_refractory(SR, None, L, fl.data_ptr(), fl.stride(0), streams, fl.shape[1], count.data_ptr(), *args, None)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
per = flags.sum(dim=1)
print("decisions per stream: max", int(per.max()), "streams with > 1000:", int((per > 1000).sum()), "median", float(per.float().median()))
cases = (("batch's decisions", flags), ("no decisions", torch.zeros_like(flags)), ("all ones", torch.ones_like(flags)),
         ("one per 100", (torch.arange(flags.shape[1], device="cuda") % 100 == 0).to(torch.uint8).expand(streams, -1).contiguous()))
for wave in (None, 1):     # the launcher's choice / a wavefront per stream for all rows      # a lane per stream (rows on 8-byte boundaries) / a wavefront per stream
    with bd.tuning(refractory_wave=wave):
        row = [f"{name}: {timed(fl):.4f} ms" for name, fl in cases]
        row.append(f"counts only: {timed(flags, False):.4f} ms")
    print(f"refractory_wave={wave} | " + " | ".join(row), flush=True)
