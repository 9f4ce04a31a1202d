"""Read-bandwidth ceiling of the GPU for the access pattern of the analysis kernels
(bd_probe_read_ex): GB/s over buffer size, wavefronts per CU, loads in flight and walk order."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd

L = bd.lib()
st = torch.cuda.current_stream().cuda_stream
out = []
for gb in (21.7, 86.7):
    n = int(gb * 1e9) // (1 << 20) * (1 << 20)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    buf.random_(0, 255)
    sink = torch.zeros(256 * 32 + 64, dtype=torch.int32, device="cuda")
    for contiguous in (0, 1):
        for depth in (4, 8, 16):
            for wpc in (2, 4, 8, 16, 32):
                ts = []
                for rep in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = L.bd_probe_read_ex(buf.data_ptr(), n, sink.data_ptr(), wpc, depth, contiguous, st)
                    e1.record()
                    torch.cuda.synchronize()
                    assert rc == 0
                    ts.append(e0.elapsed_time(e1))
                med = sorted(ts[1:])[len(ts[1:]) // 2]
                out.append({"GB": round(n / 1e9, 1), "contiguous": contiguous, "depth": depth, "waves_per_cu": wpc,
                            "ms": round(med, 3), "GBps": round(n / med / 1e6)})
                print(out[-1], flush=True)
    del buf
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "read_ceiling.json"), "w"), indent=1)
