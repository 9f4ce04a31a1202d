"""Read-bandwidth ceiling of the GPU for the access pattern of the analysis kernels
(bd_probe_read): prints GB/s for several buffer sizes and occupancies."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd

L = bd.lib()
st = torch.cuda.current_stream().cuda_stream
out = []
for gib in (21.7, 86.7):
    n = int(gib * 1e9) // 4096 * 4096
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    buf.random_(0, 255)
    sink = torch.zeros(256 * 32, dtype=torch.int32, device="cuda")
    for wpc in (8, 16, 32):
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.bd_probe_read(buf.data_ptr(), n, sink.data_ptr(), wpc, st)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            ts.append(e0.elapsed_time(e1))
        best, med = min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]
        out.append({"GB": n / 1e9, "waves_per_cu": wpc, "ms_best": best, "ms_median": med, "GBps_best": n / best / 1e6,
                    "GBps_median": n / med / 1e6})
    del buf
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
