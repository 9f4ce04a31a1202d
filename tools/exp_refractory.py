"""Time of the refractory pass on the decisions of the headline batch (tools, not product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
SR, streams, n, L = 44100, 16384, 44100 * 60, 512
buf = torch.empty((streams, (n + 63) & ~63), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=0x5EED)
res = bd.analyze_batch(bd.StrategyKind.Spectrum, SR, buf, L, n_samples=n, want_values=False)
torch.cuda.synchronize()
flags = res.flags
print("decisions set per stream:", float(flags.sum()) / streams, "beats per stream:", float(res.beat_count.sum()) / streams)
mb = res.beat_idx.shape[1]
for _ in range(3):
    bd.refractory(SR, L, flags, max_beats=mb)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L_ = bd.lib()
count, idx, ms = bd.refractory(SR, L, flags, max_beats=mb)
stride = flags.stride(0)
torch.cuda.synchronize()
a.record()
for _ in range(20):
    L_.bd_refractory(SR, None, L, flags.data_ptr(), stride, streams, flags.shape[1], count.data_ptr(), idx.data_ptr(), ms.data_ptr(), mb, None)
b.record(); torch.cuda.synchronize()
print("refractory pass: %.4f ms" % (a.elapsed_time(b) / 20), "beats", int(count.sum()))
a.record()
for _ in range(20):
    L_.bd_refractory(SR, None, L, flags.data_ptr(), stride, streams, flags.shape[1], count.data_ptr(), None, None, 0, None)
b.record(); torch.cuda.synchronize()
print("counts only:     %.4f ms" % (a.elapsed_time(b) / 20))
zero = torch.zeros_like(flags)
a.record()
for _ in range(20):
    L_.bd_refractory(SR, None, L, zero.data_ptr(), stride, streams, flags.shape[1], count.data_ptr(), idx.data_ptr(), ms.data_ptr(), mb, None)
b.record(); torch.cuda.synchronize()
print("no decisions:    %.4f ms" % (a.elapsed_time(b) / 20))
per = flags.sum(dim=1)
print("decisions per stream: max", int(per.max()), "streams with > 1000:", int((per > 1000).sum()), "median", float(per.float().median()))
for name, fl in (("all ones", torch.ones_like(flags)), ("one per 100", (torch.arange(flags.shape[1], device="cuda") % 100 == 0).to(torch.uint8).expand(streams, -1).contiguous())):
    st2 = fl.stride(0)
    L_.bd_refractory(SR, None, L, fl.data_ptr(), st2, streams, fl.shape[1], count.data_ptr(), idx.data_ptr(), ms.data_ptr(), mb, None)
    a.record()
    for _ in range(10):
        L_.bd_refractory(SR, None, L, fl.data_ptr(), st2, streams, fl.shape[1], count.data_ptr(), idx.data_ptr(), ms.data_ptr(), mb, None)
    b.record(); torch.cuda.synchronize()
    print(f"{name}: %.4f ms, beats per stream %.1f" % (a.elapsed_time(b) / 10, float(count.sum()) / streams))
