#!/usr/bin/env python3
"""Benchmark of the batched per-window analysis path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (per-callback decisions + refractory pass ->
beat lists) over one batch of synthetic streams that is already resident in
HBM.  Default workload: the Spectrum strategy on 16,384 streams x 60 s at
44.1 kHz, 512-sample callbacks (1024-point analysis window, hop 512).  With N
GPUs every rank analyses its own 16,384 streams (weak scaling, no collective on
the data path); the reported value is the audio-seconds all ranks analysed per
wall-second, timed between barriers, maximum over ranks.

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel against HBM
bandwidth using HIP events recorded inside the timed region; `cpu_baseline` is
the CPU restatement of the reference (oracle/) timed on this host's cores on a
bounded sample of the same streams (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (spec)

WORKLOADS = {
    # name: (kind, callback_len, streams, seconds, description)
    "spectrum_hop512": ("Spectrum", 512, 16384, 60,
                        "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, "
                        "1024-pt window hop 512 (512-sample callbacks), low-band max + threshold + refractory"),
    "spectrum_1024": ("Spectrum", 1024, 16384, 60,
                      "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, "
                      "1024-sample non-overlapping callbacks"),
    "lpf": ("LPF", 1024, 4096, 60,
            "LPF strategy batched: 4,096 synthetic click-track streams x 60 s @ 44.1 kHz, 1024-sample windows"),
    # callback lengths off the 64-sample grid (10 ms of audio at 44.1 kHz): other kernels behind the same entry
    "spectrum_441": ("Spectrum", 441, 16384, 60,
                     "Spectrum strategy batched: 16,384 synthetic streams x 60 s @ 44.1 kHz, 441-sample callbacks "
                     "against the 1024-sample ring"),
    "lpf_441": ("LPF", 441, 4096, 60,
                "LPF strategy batched: 4,096 synthetic click-track streams x 60 s @ 44.1 kHz, 441-sample windows"),
    # extension analysis (no reference counterpart): callback_len is the hop
    "onset_2048": ("Onset", 1024, 4096, 300,
                   "Extension mode: 4,096 synthetic 5-min tracks per GPU, 2048-pt Hann STFT hop 1024, spectral flux + "
                   "band energy + adaptive peak picking"),
    "biquad": ("Biquad", 1024, 4096, 60,
               "Extension mode: carried-state biquad envelope (120 Hz low-pass as a chunked scan), 4,096 synthetic "
               "click-track streams x 60 s @ 44.1 kHz, 1024-sample decision windows + refractory"),
    "onset_1024": ("Onset", 512, 4096, 300,
                   "Extension mode: 4,096 synthetic 5-min tracks per GPU, 1024-pt Hann STFT hop 512, spectral flux + "
                   "band energy + adaptive peak picking"),
    "onset_2048_hop512": ("Onset", 512, 4096, 300,
                          "Extension mode: 4,096 synthetic 5-min tracks per GPU, 2048-pt Hann STFT hop 512 (75 % "
                          "overlap), spectral flux + band energy + adaptive peak picking"),
}
# FFT length of the onset workloads (callback_len is their hop)
ONSET_FFT = {"onset_2048": 2048, "onset_1024": 1024, "onset_2048_hop512": 2048}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="spectrum_hop512", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU (testing only)")
    ap.add_argument("--seconds", type=int, default=0, help="override stream length in seconds (testing only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-streams", type=int, default=1024, help="streams in the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / timing reduction (gloo: rehearsals)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (a box with one GPU)")
    args = ap.parse_args()

    import torch
    import beat_detector_amd as bd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    kind_name, L, streams, seconds, desc = WORKLOADS[args.workload]
    if args.streams:
        streams = args.streams
    if args.seconds:
        seconds = args.seconds
    onset = kind_name == "Onset"
    biquad = kind_name == "Biquad"
    kind = None if onset else bd.StrategyKind["LPF" if biquad else kind_name]
    sr = 44100
    n = sr * seconds
    stride = (n + 63) & ~63  # rows start on 128-byte boundaries
    seed = 0x5EED
    # weak scaling: `streams` per GPU, rank r owns the r-th contiguous shard of world * streams
    from beat_detector_amd import sharding
    first_stream = sharding.shard_range(world * streams, rank, world)[0]

    samples = torch.empty((streams, stride), dtype=torch.int16, device=dev)
    bd.synth_fill(samples, n_samples=n, first_stream=first_stream, seed=seed, sampling_rate=sr)
    nfft = ONSET_FFT.get(args.workload, 0)
    if onset:
        oparams = bd.OnsetParams.default(fft_len=nfft, hop=L)
        n_cb = bd.onset_frames(n, nfft, L)               # frames per track
        max_beats = max(1, (n * 1000 // sr) // max(oparams.min_gap_ms, 1) + 1)
    else:
        n_cb = n // L
        max_beats = max(1, bd.batch_max_beats(kind, sr, n, L))
    # output rows start on 128-byte boundaries so that every store is a full line
    out_stride = (n_cb + 127) & ~127
    values = torch.empty((streams, out_stride), dtype=torch.float32, device=dev)
    flags = torch.empty((streams, out_stride), dtype=torch.float32 if onset else torch.uint8, device=dev)
    count = torch.zeros((streams,), dtype=torch.int32, device=dev)
    idx = torch.zeros((streams, max_beats), dtype=torch.int32, device=dev)
    ms = torch.zeros((streams, max_beats), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    L_ = bd.lib()
    st = torch.cuda.current_stream().cuda_stream

    import ctypes

    def stage1_onset():
        # `values` holds the flux rows, `flags` the band-energy rows
        rc = L_.bd_onset_stft(sr, ctypes.byref(oparams), samples.data_ptr(), streams, n, stride, values.data_ptr(),
                              flags.data_ptr(), out_stride, st)
        if rc != 0:
            raise RuntimeError(L_.bd_last_error().decode())

    def stage2_onset():
        rc = L_.bd_onset_pick(sr, ctypes.byref(oparams), values.data_ptr(), streams, n_cb, out_stride, None,
                              count.data_ptr(), idx.data_ptr(), ms.data_ptr(), max_beats, st)
        if rc != 0:
            raise RuntimeError(L_.bd_last_error().decode())

    def stage1():
        rc = L_.bd_batch_candidates(kind.value, sr, None, samples.data_ptr(), streams, n, stride, L,
                                    values.data_ptr(), flags.data_ptr(), out_stride, st)
        if rc != 0:
            raise RuntimeError(L_.bd_last_error().decode())

    def stage2():
        rc = L_.bd_refractory(sr, None, L, flags.data_ptr(), out_stride, streams, n_cb, count.data_ptr(), idx.data_ptr(),
                              ms.data_ptr(), max_beats, st)
        if rc != 0:
            raise RuntimeError(L_.bd_last_error().decode())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    bq_params = bd.BiquadParams.default(window_len=L) if biquad else None
    bq_design = bd.biquad_design(sr, bq_params.cutoff_hz, bq_params.q_factor) if biquad else None

    def stage1_biquad():
        rc = L_.bd_biquad_envelope(sr, ctypes.byref(bq_params), ctypes.byref(bq_design), samples.data_ptr(), streams, n,
                                   stride, values.data_ptr(), flags.data_ptr(), out_stride, st)
        if rc != 0:
            raise RuntimeError(L_.bd_last_error().decode())

    if onset:
        stage1, stage2 = stage1_onset, stage2_onset
    if biquad:
        stage1 = stage1_biquad
    stage1()
    launched = bd.last_kernel_name()      # the kernel the entry point routed this workload to
    stage2()
    for _ in range(args.warmup):
        stage1()
        stage2()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        stage1()                      # dominant kernel
        ev[k][1].record()
        stage2()                      # refractory pass
        ev[k][2].record()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if os.environ.get("BENCH_VERBOSE"):
        print("per-step kernel ms:", [round(e[0].elapsed_time(e[1]), 3) for e in ev], file=sys.stderr)
    k1_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    k2_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    kernel_name = launched or {"Spectrum": "bd_spectrum_band_kernel", "LPF": "bd_lpf_window_kernel",
                               "Onset": "bd_stft_flux_kernel", "Biquad": "bd_biquad_env_kernel"}[kind_name]
    total_beats = int(count.sum().item())

    if dist is not None:
        t = torch.tensor([elapsed, k1_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, k1_ms = float(t[0]), float(t[1])
        tb = torch.tensor([total_beats], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tb)
        total_beats = int(tb[0])

    audio_sec_per_step = world * streams * (n / sr)
    value = audio_sec_per_step * args.steps / elapsed
    # algorithmic bytes of one launch of the dominant kernel: every sample that
    # belongs to a callback read once (2 B) + 5 B written per callback
    alg_bytes = streams * (n_cb * L * 2 + n_cb * 5)
    if onset:  # every sample of a complete frame once + flux and band energy (8 B) per frame
        alg_bytes = streams * (((n_cb - 1) * L + nfft) * 2 + n_cb * 8) if n_cb else 0
    achieved = alg_bytes / (k1_ms * 1e-3) / 1e9

    result = {
        "metric": "audio-sec analysed/sec (node), " + {"Spectrum": "1024-pt STFT", "Onset": "%d-pt STFT" % nfft,
                                                       "LPF": "lowpass windows",
                                                       "Biquad": "carried-state biquad"}[kind_name],
        "value": value,
        "unit": "audio-sec/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": desc,
            "strategy": kind_name,
            "streams_per_gpu": streams,
            "seconds_per_stream": seconds,
            "sampling_rate": sr,
            "callback_len": L,
            "callbacks_per_stream": n_cb,
            "seed": seed,
            "parallelism": f"stream-sharded x{world}, no collectives",
            "beats_found": total_beats,
        },
        "roofline": {
            "bound": "hbm",
            "note": ("this kernel is limited by f32 VALU issue (a 2048-point FFT per 1024 new samples), not by HBM; "
                     "the HBM fraction is reported for comparison with the other workloads") if onset else None,
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": _pmc_traffic(args.workload, streams, seconds),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": k1_ms,
            "refractory_ms": k2_ms,
        },
    }

    if onset:
        # a full FFT per frame is bounded by f32 vector throughput as well: 2.5 N log2 N flops per real FFT
        # against the packed-FMA vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 x 2 at 2.4 GHz)
        flops = streams * n_cb * 2.5 * nfft * (nfft.bit_length() - 1)
        peak_tflops = 256 * 4 * 16 * 2 * 2 * 2.4e9 / 1e12
        result["roofline"]["valu"] = {"achieved": flops / (k1_ms * 1e-3) / 1e12, "peak": peak_tflops, "unit": "TFLOP/s",
                                      "frac": flops / (k1_ms * 1e-3) / 1e12 / peak_tflops,
                                      "flops_per_frame": 2.5 * nfft * (nfft.bit_length() - 1)}
    if result["roofline"].get("note") is None:
        result["roofline"].pop("note", None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(bd, kind_name, samples, n, L, sr, min(args.cpu_streams, streams), nfft=nfft)
        # the GPU's count over the same streams (equal counts are a full-size consistency check, not the parity test)
        cb["gpu_events_same_sample"] = int(count[:cb.pop("_n")].sum().item())
        result["cpu_baseline"] = cb
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic(workload, streams, seconds):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    PMC passes of this same command (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t.get(workload)
        if e and e.get("streams") == streams and e.get("seconds") == seconds:
            return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def cpu_baseline(bd, kind_name, samples, n, L, sr, n_streams, nfft=0):
    """The CPU restatement of the reference (oracle/, a port: the Rust original
    cannot be built here) on the first `n_streams` streams of the same batch."""
    if kind_name == "Onset":
        return cpu_baseline_onset(samples, n, L, nfft, sr, min(n_streams, 128))
    if kind_name == "Biquad":
        return cpu_baseline_biquad(samples, n, L, sr, min(n_streams, 512))
    from oracle import bd_oracle
    okind = bd_oracle.KIND_SPECTRUM if kind_name == "Spectrum" else bd_oracle.KIND_LPF
    host = samples[:n_streams, :n].cpu().numpy()
    cores = 16
    try:
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        pass
    t0 = time.perf_counter()
    total, _ = bd_oracle.run_batch(okind, sr, host, L, n_threads=cores)
    dt = time.perf_counter() - t0
    one = max(1, n_streams // (cores * 2))
    t1 = time.perf_counter()
    bd_oracle.run_batch(okind, sr, host[:one], L, n_threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": n_streams * (n / sr) / dt,
        "unit": "audio-sec/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n_streams} streams x {n // sr} s of the same batch, {cores} threads "
                  f"(one contiguous stream shard each), {dt:.1f} s of wall time; beats found {total}",
        "cpu_events": int(total),
        "_n": n_streams,
        "value_1core": one * (n / sr) / dt1,
        "sample_1core": f"first {one} streams, 1 thread, {dt1:.1f} s",
    }


def cpu_baseline_biquad(samples, n, L, sr, n_streams):
    """The CPU definition of the carried-state biquad envelope (oracle/bd_oracle_ext.c)
    plus the refractory replay, one stream per task."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import bd_oracle, bd_oracle_ext as ox
    host = samples[:n_streams, :n].cpu().numpy()
    cores = 16
    try:
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        pass
    p = ox.biquad_params(window_len=L)
    d = ox.biquad_design(sr, p.cutoff_hz, p.q_factor)
    ol = bd_oracle.lib()

    def one(row):
        _, flags = ox.biquad_envelope(row, sr, p, d)
        st = bd_oracle.State()
        ol.bdo_state_new(st, sr)
        beats = 0
        for f in flags:
            ol.bdo_state_update_time(st, L)
            if f and ol.bdo_last_beat_beyond_threshold(st, 400):
                ol.bdo_state_mark_beat(st)
                beats += 1
        return beats

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(one, host))
    dt = time.perf_counter() - t0
    return {
        "value": n_streams * (n / sr) / dt,
        "unit": "audio-sec/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n_streams} streams x {n // sr} s of the same batch, {cores} threads, {dt:.1f} s of wall "
                  f"time; beats found {total} (the definition of the extension analysis; refractory replay in Python)",
        "cpu_events": int(total),
        "_n": n_streams,
    }


def cpu_baseline_onset(samples, n, hop, nfft, sr, n_streams):
    """The CPU definition of the extension analysis (oracle/bd_oracle_ext.c) on the
    first `n_streams` tracks; ctypes releases the GIL, one track per task."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import bd_oracle_ext as ox
    host = samples[:n_streams, :n].cpu().numpy()
    cores = 16
    try:
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        pass
    p = ox.default_params(fft_len=nfft, hop=hop)

    def one(row):
        flux, _ = ox.onset_stft(row, sr, p)
        return len(ox.onset_pick(flux, sr, p)[1])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(one, host))
    dt = time.perf_counter() - t0
    return {
        "value": n_streams * (n / sr) / dt,
        "unit": "audio-sec/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n_streams} tracks x {n // sr} s of the same batch, {cores} threads, {dt:.1f} s of wall "
                  f"time; onsets found {total} (scalar radix-2 FFT: the definition of the analysis, not a tuned CPU FFT)",
        "cpu_events": int(total),
        "_n": n_streams,
    }


if __name__ == "__main__":
    main()
