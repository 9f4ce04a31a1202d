"""LPF off the 64-sample grid: the line-staging kernels side by side in one process (BD_LPF_VARIANT 6 = per-piece
64-bit addressing, 5 = 32-bit offsets through a buffer descriptor, default = the same with the shared line kept in
registers; a suffix gN sets BD_LPF_GROUPS=N).  Environment: STREAMS, LENGTHS, ROUNDS, VARIANTS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beat_detector_amd as bd
sr = 44100; streams = int(os.environ.get("STREAMS", "4096")); n = sr * 60; stride = (n + 63) & ~63
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
variants = os.environ.get("VARIANTS", "6,5,0").split(",")
for L in [int(v) for v in os.environ.get("LENGTHS", "441,500,529,540,882,1000,1500,300,257").split(",")]:
    out = {v: [] for v in variants}
    sums = {}
    n_cb = n // L; pitch = (n_cb + 127) & ~127
    vals = torch.empty((streams, pitch), dtype=torch.float32, device="cuda")
    flags = torch.empty((streams, pitch), dtype=torch.uint8, device="cuda")
    for r in range(int(os.environ.get("ROUNDS", "3"))):
        for v in variants:
            os.environ["BD_LPF_VARIANT"], _, groups = v.partition("g")      # "0g4": variant 0, BD_LPF_GROUPS=4
            os.environ.pop("BD_LPF_GROUPS", None)
            if groups:
                os.environ["BD_LPF_GROUPS"] = groups
            bd.lib().bd_tuning_reload()
            vals.zero_(); flags.zero_()
            res = bd.batch_candidates(bd.StrategyKind.LPF, sr, buf, L, n_samples=n, values=vals, flags=flags)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                res = bd.batch_candidates(bd.StrategyKind.LPF, sr, buf, L, n_samples=n, values=vals, flags=flags)
            b.record(); torch.cuda.synchronize()
            out[v].append(a.elapsed_time(b) / 4)
            sums[v] = (int(flags[:, :n_cb].sum()), float(vals[:, :n_cb].double().sum()))
    print(f"L={L:5d} " + " | ".join(f"v{v}: " + " ".join(f"{t:.3f}" for t in ts) + f" ms (best {streams * (n // L) * L * 2 / min(ts) / 1e9:.2f} TB/s)"
                                    for v, ts in out.items()) + f"  decisions {sums[variants[0]][0]} all equal: {len(set(sums.values())) == 1}", flush=True)
