"""Two builds of the library in one process, on the same buffers (tools, not product): the launch time of a kernel
moves by about 3 % from process to process with the placement of the sample buffer, so builds are compared here,
alternating, inside one process.  Usage: exp_ab.py <other libbeatdet_hip*.so>   (environment: STREAMS, LENGTHS, KIND = Spectrum | Lpf | biquad; for biquad
LENGTHS are the decision windows)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd

SR = 44100
streams, n = int(os.environ.get("STREAMS", "16384")), SR * 60
biquad = os.environ.get("KIND", "Spectrum") == "biquad"
kind = None if biquad else bd.StrategyKind[os.environ.get("KIND", "Spectrum")]
libs = {"product": bd.lib(), "other": C.CDLL(os.path.abspath(sys.argv[1]))}
vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
libs["other"].bd_batch_candidates.restype = i32
libs["other"].bd_batch_candidates.argtypes = [i32, u32, vp, vp, u64, u64, u64, u32, vp, vp, u64, vp]
libs["other"].bd_biquad_envelope.restype = i32
libs["other"].bd_biquad_envelope.argtypes = [u32, vp, vp, vp, u64, u64, u64, vp, vp, u64, vp]
buf = torch.empty((streams, (n + 63) & ~63), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=3)
st = torch.cuda.current_stream().cuda_stream
for L in [int(v) for v in os.environ.get("LENGTHS", "441,529").split(",")]:
    n_cb = n // L
    pitch = (n_cb + 127) & ~127
    vals = torch.empty((streams, pitch), dtype=torch.float32, device="cuda")
    flags = torch.empty((streams, pitch), dtype=torch.uint8, device="cuda")
    ref = None
    out = {k: [] for k in libs}
    for r in range(int(os.environ.get("ROUNDS", "3"))):
        for name, lib in libs.items():
            def go():
                if biquad:
                    prm = bd.api.BiquadParams.default()
                    prm.window_len = L
                    rc = lib.bd_biquad_envelope(SR, C.byref(prm), None, buf.data_ptr(), streams, n,
                                                buf.shape[1], vals.data_ptr(), flags.data_ptr(), pitch, st)
                    assert rc == 0
                    return
                rc = lib.bd_batch_candidates(kind.value, SR, None, buf.data_ptr(), streams, n, buf.shape[1], L, vals.data_ptr(),
                                             flags.data_ptr(), pitch, st)
                assert rc == 0
            go()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                go()
            b.record()
            torch.cuda.synchronize()
            out[name].append(a.elapsed_time(b) / 4)
            total = int(flags[:, :n_cb].sum())
            ref = total if ref is None else ref
            assert total == ref or os.environ.get("ALLOW_DIFFERENT"), (name, total, ref)       # both builds decide alike
    print(f"L={L:5d} " + " | ".join(f"{k}: " + " ".join(f"{t:.3f}" for t in v) + f" ms (best {streams*n_cb*L*2/min(v)/1e9:.2f} TB/s)"
                                    for k, v in out.items()) + f"  decisions {ref}", flush=True)
