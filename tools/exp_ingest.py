"""Throughput of the ingest converters (u16 -> i16, f32 -> i16, stereo -> mono) on 2 Gi samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
n = 1 << 31
def timed(fn, bytes_moved):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    return round(ms, 3), round(bytes_moved / ms / 1e6)
u = torch.randint(-32768, 32767, (n,), dtype=torch.int16, device="cuda")
print("u16->i16 (ms, GB/s read+write)", timed(lambda: bd.u16_data_to_i16(u), n * 4))
print("stereo->mono", timed(lambda: bd.stereo_to_mono(u), n * 2 + n))
del u
f = torch.rand((n,), dtype=torch.float32, device="cuda") * 2 - 1
print("f32->i16", timed(lambda: bd.f32_data_to_i16(f), n * 6))
