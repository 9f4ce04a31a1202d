"""Rate of the on-device synthetic generator and of a chunked pass (tools, not product)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import beat_detector_amd as bd
SR = 44100
n = SR * 300
buf = torch.empty((4096, (n + 63) & ~63), dtype=torch.int16, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bd.synth_fill(buf, n_samples=n, seed=0x5EED)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"synth 4096 x 300 s: {dt*1e3:.1f} ms  {buf.numel()*2/dt/1e12:.2f} TB/s written")
for s in (0, 1777, 4095):
    want = bd.synth_fill_host(1, n, first_stream=s, seed=0x5EED)[0]
    assert np.array_equal(buf[s, :n].cpu().numpy(), want), s
print("bit-identical to the host generator on sampled streams")
